#!/bin/bash
# Round-end evidence (run on the GPU box from the repo root): the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, and the separate PMC passes behind profiles/r01_traffic.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 500 python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
echo bench done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -- python3 bench.py --no-cpu-baseline > gpurun_out/final/bench_under_rocprof.json 2> gpurun_out/final/kt.err
echo ktrace done
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "hit TCC_HIT_sum" "miss TCC_MISS_sum"; do
  set -- $pass
  for rend in wavefront megakernel; do
    timeout -k 5 150 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/final/pmc_${rend}_$1 -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-pipelined --renderer $rend > gpurun_out/final/pmc_${rend}_$1.log 2>&1
    echo "pmc $rend $1 rc=$?"
  done
done
