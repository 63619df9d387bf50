#!/bin/bash
# A/B timing of kernel variants on the GPU box: scripts/ab.sh SPP [mega|wave|both] name1 name2 ...   ("main" = the in-tree library)
spp=$1; shift; which=$1; shift
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = main ]; then
    env -u RT_MI355X_LIB python3 scripts/quick_bench.py $spp $which 2>&1 | grep -v "^\[rt stats\] waves\|timeline"
  else
    RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so python3 scripts/quick_bench.py $spp $which 2>&1 | grep -v "^\[rt stats\] waves\|timeline"
  fi
done
