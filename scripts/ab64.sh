#!/bin/bash
# scripts/ab64.sh name1 name2 ...: the bench configuration (atrium 1080p, 64 spp), both renderers, each variant TWICE in alternation (run-to-run noise is ~0.5 %)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  echo "=== $v (run $rep)"
  python3 scripts/quick_bench.py 64 both 2>&1 | grep -v "^\[rt stats\]"
done
done
