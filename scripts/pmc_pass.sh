#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <renderer> <counters...>   (run on the GPU box from the repo root)
tag=$1; shift; rend=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-pipelined --renderer $rend > gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag rc=$?"
