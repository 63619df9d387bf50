#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b3
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/b3/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/b3/gpu_tests.log
tail -4 gpurun_out/b3/gpu_tests.log
grep -q "rc=0" gpurun_out/b3/gpu_tests.log || exit 1
timeout -k 10 600 scripts/ab64.sh nopack noofs main > gpurun_out/b3/ab64.log 2>&1 && cat gpurun_out/b3/ab64.log
