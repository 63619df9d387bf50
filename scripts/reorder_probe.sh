#!/bin/bash
# f-3 measurement on the GPU box: wavefront renderer, atrium 1080p, 8 spp, default order vs RT_WF_REORDER / RT_WF_MATSORT:
# Mrays/s, EXTEND lanes per step (RT_KERNEL_STATS=1) and the L2 hit rate of k_wf_extend / k_wf_shade (rocprofv3 --pmc). -> gpurun_out/r02/reorder/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02/reorder; mkdir -p $out
for cfg in "base 0 0" "reorder 1 0" "matsort 0 1" "both 1 1"; do
  set -- $cfg
  export RT_WF_REORDER=$2 RT_WF_MATSORT=$3
  python3 scripts/quick_bench.py 8 wave > $out/$1.bench.log 2>&1
  RT_KERNEL_STATS=1 python3 scripts/quick_bench.py 8 wave > $out/$1.stats.log 2>&1
  timeout -k 5 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/$1.pmc -- python3 scripts/quick_bench.py 2 wave > $out/$1.pmc.log 2>&1
  echo "$1: $(grep Wavefront $out/$1.bench.log)"
done
