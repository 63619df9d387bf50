#!/bin/bash
# Round-end evidence (run on the GPU box from the repo root; outputs under gpurun_out/r02/final, copied into profiles/ by scripts/collect_profiles.py):
#   1. the VALU calibration with the CURRENT kernels' instruction mixes (scripts/gen_valu_mix.py must have been run on the current build)
#   2. the default bench line (which measures traffic + the VALU-issue bound itself, in child processes under rocprofv3 --pmc)
#   3. the same command under rocprofv3 --kernel-trace --stats (average launch duration of the dominant kernels)
#   4. PMC passes of one frame per renderer at the bench configuration (the fallback file bench.py reads when rocprofv3 is not usable)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02/final; mkdir -p $out
bash scripts/valu_calib.sh > $out/calib.log 2>&1
echo calib done
RT_KERNEL_STATS=1 python3 scripts/quick_bench.py 8 mega > $out/stats.log 2>&1
python3 scripts/config_table.py > $out/config_table.txt 2>&1
python3 scripts/tile_scaling.py 64 "" 1,2,4,8 > $out/tile_scaling.txt 2>&1
echo stats done
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err
echo bench done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --no-live-pmc > $out/bench_under_rocprof.json 2> $out/kt.err
echo ktrace done
for rend in megakernel wavefront; do
  for pass in "fetch FETCH_SIZE TCC_HIT_sum" "write WRITE_SIZE TCC_MISS_sum" "valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "wave SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU" "lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
    set -- $pass; tag=$1; shift
    timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_${rend}_$tag -- python3 bench.py --pmc-child --renderer $rend > $out/pmc_${rend}_$tag.log 2>&1
    echo "pmc $rend $tag rc=$?"
  done
done
