#!/bin/bash
# Round 3 calibration on the GPU box (run from the repo root): scripts/valu_calib2.hip alone, then under the SQ counters whose meaning
# it pins down (one pass per counter group, counters only). scripts/valu_calib2_summary.py turns the outputs into
# profiles/r03_valu_calibration2.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/calib2
rm -rf $O; mkdir -p $O scripts/build
[ -x scripts/build/valu_calib2 ] || hipcc -O2 -std=c++17 --offload-arch=gfx950 scripts/valu_calib2.hip -o scripts/build/valu_calib2
timeout -k 10 120 scripts/build/valu_calib2 4000 > $O/plain.json
echo "plain done"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_busy -- scripts/build/valu_calib2 4000 > $O/under_busy.json 2> $O/pmc_busy.err
echo "busy pass done"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/pmc_types1 -- scripts/build/valu_calib2 4000 > $O/under_types1.json 2> $O/pmc_types1.err
echo "types1 pass done"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_types2 -- scripts/build/valu_calib2 4000 > $O/under_types2.json 2> $O/pmc_types2.err
echo "types2 pass done"
find $O -name "*counter_collection.csv" | head
