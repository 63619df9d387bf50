#!/bin/bash
# VALU issue-ceiling calibration on the GPU box (run from the repo root): the microbenchmark alone, then the same binary under the
# PMC counters used on the render kernels. Outputs under gpurun_out/r02/; scripts/valu_calib_summary.py turns them into profiles/r02_valu_calibration.json
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02 scripts/build
[ -x scripts/build/valu_calib ] || hipcc -O2 --offload-arch=gfx950 scripts/valu_calib.hip -o scripts/build/valu_calib
timeout -k 10 200 scripts/build/valu_calib 4000 > gpurun_out/r02/valu_calib.json
echo "calib plain done"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r02/valu_calib_pmc -- scripts/build/valu_calib 4000 > gpurun_out/r02/valu_calib_under_pmc.json 2> gpurun_out/r02/valu_calib_pmc.err
echo "calib pmc done"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/r02/valu_calib_pmc2 -- scripts/build/valu_calib 4000 > gpurun_out/r02/valu_calib_under_pmc2.json 2> gpurun_out/r02/valu_calib_pmc2.err
echo "calib pmc2 done"
