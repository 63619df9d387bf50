#!/bin/bash
# wavefront renderer by frame size (depth 10, 16 spp), default settings, per library variant
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  for res in 1920x540 1920x1080 3840x2160; do
    echo "== $v $res: $(RES=$res python3 scripts/quick_bench.py 16 wave 2>&1 | grep Wavefront)"
  done
done
