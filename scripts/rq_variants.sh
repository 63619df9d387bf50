#!/bin/bash
# requeue variants: 1080p and 1920x540 at C = 16, depth 10, 16 spp
cd $GRAFT_REPO_ROOT
for v in main "$@"; do
  if [ $v = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  for res in 1920x540 1920x1080; do
    echo "== $v $res C=16: $(RES=$res RT_WF_SAMPLES_PER_LAUNCH=16 python3 scripts/quick_bench.py 16 wave 2>&1 | grep Wavefront)"
  done
done
