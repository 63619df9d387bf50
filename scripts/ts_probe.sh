#!/bin/bash
# k_megakernel_ts (RT_MEGA_TS=1) against k_megakernel: full frame at 16 spp, then the wave statistics of an 8 spp frame, per library variant
cd $GRAFT_REPO_ROOT
echo "== k_megakernel";   python3 scripts/quick_bench.py 16 mega 2>&1 | grep Mega
for v in main "$@"; do
  if [ $v = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  echo "== k_megakernel_ts $v"; RT_MEGA_TS=1 python3 scripts/quick_bench.py 16 mega 2>&1 | grep Mega
  RT_MEGA_TS=1 RT_KERNEL_STATS=1 python3 scripts/quick_bench.py 8 mega 2>&1 | grep "rt stats" | tail -2
done
