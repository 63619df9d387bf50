#!/bin/bash
# usage: scripts/ktrace.sh <tag> <quick_bench args...> : per-kernel time summary via rocprofv3 --kernel-trace --stats
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python3 scripts/quick_bench.py "$@" > gpurun_out/kt_$tag.log 2>&1
cat gpurun_out/kt_$tag/*/*_kernel_stats.csv | cut -c1-60,120-400 | grep -v "at::native" | head -12
grep Mrays gpurun_out/kt_$tag.log
