#!/bin/bash
# Round 4, item 1: what binds the CHAIN REGIME? rocprofv3 --pmc passes (counters only) of the megakernel on 1/8 of the 1080p / 64 spp
# frame (259,200 pixels < 393,216 lanes) and, for comparison, on the full frame.
#   usage (GPU box, repo root): bash scripts/r04/tile_pmc.sh <tag> [renderer=mega] [worlds="8 1"]
set -e
TAG=${1:-tile}; R=${2:-mega}; WORLDS=${3:-"8 1"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r04/pmc_$TAG
mkdir -p $O
for WORLD in $WORLDS; do
  i=0
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $O/w${WORLD}_p$i -- python3 scripts/tile_scaling.py 64 $R $WORLD > $O/w${WORLD}_p$i.out 2> $O/w${WORLD}_p$i.err || { echo "pass $i (1/$WORLD) failed"; tail -3 $O/w${WORLD}_p$i.err; }
  done
done
python3 scripts/r04/tile_pmc_summary.py $O $WORLDS | tee $O/summary.txt
