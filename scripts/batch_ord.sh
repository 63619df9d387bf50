#!/bin/bash
# one GPU call: parity of the ordered-children variant, A/B at the bench configuration, steps per ray, both workload classes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ord
timeout -k 10 600 python3 -m pytest tests/test_gpu_experiments.py -x -q -k "ordered" > gpurun_out/ord/parity.log 2>&1 || { tail -30 gpurun_out/ord/parity.log; exit 1; }
tail -3 gpurun_out/ord/parity.log
timeout -k 10 400 scripts/ab64.sh base main ord > gpurun_out/ord/ab64.log 2>&1 && cat gpurun_out/ord/ab64.log
for v in main ord; do
  if [ "$v" = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  echo "=== stats $v"
  RT_KERNEL_STATS=1 timeout -k 10 200 python3 scripts/quick_bench.py 4 mega 2>&1 | grep "rt stats\] rays" | tail -1
done > gpurun_out/ord/stats.log 2>&1
cat gpurun_out/ord/stats.log
unset RT_MI355X_LIB
timeout -k 10 400 scripts/ab_scenes.sh main ord > gpurun_out/ord/scenes.log 2>&1 && cat gpurun_out/ord/scenes.log
