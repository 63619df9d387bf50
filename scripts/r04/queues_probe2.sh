#!/bin/bash
cd $GRAFT_REPO_ROOT
for Q in 8 16; do
  for FD in 0 1; do
    echo "== auto GPU_MAX_HW_QUEUES=$Q RT_BENCH_FORCE_DIST=$FD"
    GPU_MAX_HW_QUEUES=$Q RT_BENCH_FORCE_DIST=$FD python3 bench.py --steps 3 --warmup 1 --no-live-pmc --no-cpu-baseline --no-pipelined 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(v['value'],v['ms_per_step']) for k,v in d['renderers'].items()})"
  done
done
