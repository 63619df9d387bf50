#!/bin/bash
# Round-3 evidence (GPU box, repo root; outputs under gpurun_out/r03/final, copied into profiles/ by scripts/collect_profiles_r03.py):
#   1. the default bench line as the driver runs it (bench.py measures traffic + VALU busy itself, in child processes under rocprofv3 --pmc)
#   2. the same command under rocprofv3 --kernel-trace --stats (average launch duration of the dominant kernels; no PMC, no CPU leg)
#   3. the wavefront renderer's schedules side by side at 64 spp, BASELINE's configurations, tile scaling
#   4. wave-level step statistics of the final kernels (RT_KERNEL_STATS=1)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03/final; mkdir -p $out
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --dump-pmc $out/pmc_bench.json > $out/bench.json 2> $out/bench.err
echo bench done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --no-pipelined > $out/bench_under_rocprof.json 2> $out/kt.err
echo ktrace done
python3 scripts/schedule_probe.py 64 > $out/schedules_64spp.txt 2>&1
echo schedules done
python3 scripts/config_table.py > $out/config_table.txt 2>&1
python3 scripts/tile_scaling.py 64 "" 1,2,4,8 > $out/tile_scaling.txt 2>&1
echo tables done
RT_KERNEL_STATS=1 python3 scripts/quick_bench.py 8 both > $out/stats.log 2>&1
echo stats done
