#!/bin/bash
# scripts/ab_lib.sh "<renderer> <worlds> <label:ENV=..>..." name1 name2 ...: tile_ab.py for library variants (scripts/build_variant.sh), each
# twice in alternation inside ONE GPU call (boxes differ by ~1 %). "main" = the product library.
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  echo "=== $v (run $rep)"
  python3 scripts/tile_ab.py $ARGS 2>&1 | grep -v "^\[rt stats\]"
done
done
