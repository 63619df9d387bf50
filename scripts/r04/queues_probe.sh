#!/bin/bash
# How many hardware queues does the per-bounce pipeline need next to torch + RCCL? (bench.py, split EXTEND + SHADE on three stream lanes)
cd $GRAFT_REPO_ROOT
for Q in 8 12 16 24; do
  for FD in 0 1; do
    echo "== GPU_MAX_HW_QUEUES=$Q RT_BENCH_FORCE_DIST=$FD"
    GPU_MAX_HW_QUEUES=$Q RT_BENCH_FORCE_DIST=$FD python3 bench.py --steps 3 --warmup 1 --no-live-pmc --no-cpu-baseline --no-pipelined --renderer wavefront_per_bounce 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['renderers']['wavefront_per_bounce']; print(r['value'], r['ms_per_step'], r['roofline'].get('pipeline',{}).get('stream_lanes'))"
  done
done
