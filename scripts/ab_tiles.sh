#!/bin/bash
# scripts/ab_tiles.sh SPP WORLDS name1 name2 ... : scripts/tile_scaling.py (megakernel) for kernel variants ("main" = in-tree library)
spp=$1; shift; worlds=$1; shift
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = main ]; then env -u RT_MI355X_LIB python3 scripts/tile_scaling.py $spp mega $worlds 2>&1
  else RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so python3 scripts/tile_scaling.py $spp mega $worlds 2>&1; fi
done
