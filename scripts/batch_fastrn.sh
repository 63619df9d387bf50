#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fastrn
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/fastrn/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/fastrn/gpu_tests.log
tail -5 gpurun_out/fastrn/gpu_tests.log
grep -q "rc=0" gpurun_out/fastrn/gpu_tests.log || exit 1
timeout -k 10 600 scripts/ab64.sh nofast main nt sh70 sh80 > gpurun_out/fastrn/ab64.log 2>&1 && cat gpurun_out/fastrn/ab64.log
