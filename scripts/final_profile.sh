#!/bin/bash
# Everything committed under profiles/r05_* (GPU box, repo root; outputs under gpurun_out/r05/final, copied into profiles/ by
# `python3 scripts/final_profile.sh collect` ... see the end of this file):
#   1. the default bench line as the driver runs it (bench.py measures traffic + VALU busy itself, child processes under rocprofv3 --pmc)
#   2. the same command under rocprofv3 --kernel-trace --stats (mean launch duration of the dominant kernels; no PMC, no CPU leg)
#   3. the wavefront renderer's schedules at 64 spp, BASELINE's configurations, tile scaling (both renderers)
#   4. counters of the megakernel on 1/8 of the frame against the full frame (what binds the chain regime)
#   5. wave-level step statistics and frame timelines (RT_KERNEL_STATS=1), one frame and two frames in flight
#   6. pixel slices on 1/N tiles (both renderers), the chain floor (ever smaller tiles), the per-bounce kernel's rate against rays per launch
# usage: bash scripts/final_profile.sh [part ...]   (parts: bench ktrace tables tilepmc stats misc; default: all)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r05/final; mkdir -p $out
parts=${@:-bench ktrace tables tilepmc stats misc}
for part in $parts; do case $part in
bench)
  timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --dump-pmc $out/pmc_bench.json > $out/bench.json 2> $out/bench.err
  echo bench done ;;
ktrace)
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --no-pipelined --no-sensitivity > $out/bench_under_rocprof.json 2> $out/kt.err
  echo ktrace done ;;
tables)
  timeout -k 10 400 python3 scripts/schedule_probe.py 64 > $out/schedules_64spp.txt 2>&1
  timeout -k 10 400 python3 scripts/config_table.py > $out/config_table.txt 2>&1
  timeout -k 10 400 python3 scripts/tile_scaling.py 64 "" 1,2,4,8 1,2 > $out/tile_scaling.txt 2>&1
  echo tables done ;;
tilepmc)
  bash scripts/tile_pmc.sh final mega "8 1" > $out/tile_pmc.txt 2>&1
  echo tilepmc done ;;
stats)
  (echo "== RT_KERNEL_STATS=2: the timing-only instantiation (cycles per step / per shading round: the product's within ~1 %)"; RT_KERNEL_STATS=2 python3 scripts/tile_scaling.py 64 mega 1,8 1 2>&1 | grep -E "rt stats|Renderer" | tail -12; echo "== RT_KERNEL_STATS=1: the fully instrumented instantiation (lanes per step, node visits, frame timeline; its cycles read ~5-30 % long)"; RT_KERNEL_STATS=1 python3 scripts/tile_scaling.py 64 mega 1,8 1 2>&1 | grep -E "rt stats|Renderer" | tail -16) > $out/stats.log
  RT_KERNEL_STATS=1 python3 bench.py --steps 2 --warmup 1 --renderer megakernel --no-cpu-baseline --no-live-pmc --no-sensitivity 2>&1 | grep -E "frame timeline" | tail -6 >> $out/stats.log
  echo stats done ;;
misc)
  timeout -k 10 400 python3 scripts/slice_probe.py 1,2,4,8 0,-1 64 mega > $out/pixel_slices.txt 2>&1
  timeout -k 10 400 python3 scripts/slice_probe.py 1,2,4,8 0,-1 64 wave >> $out/pixel_slices.txt 2>&1
  timeout -k 10 400 python3 scripts/tile_floor.py 64 8,16,32,64 > $out/tile_floor.txt 2>&1
  timeout -k 10 400 python3 scripts/shoot_rate.py > $out/shoot_rate.txt 2>&1
  echo misc done ;;
esac; done
