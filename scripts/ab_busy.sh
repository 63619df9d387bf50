#!/bin/bash
# scripts/ab_busy.sh SPP RENDERER name1 name2 ...: A/B of kernel variants (scripts/build_variant.sh; "main" = the product library) on the GPU box:
# frame time (scripts/quick_bench.py) and the measured VALU busy of the dominant kernel (one rocprofv3 --pmc pass: SQ_ACTIVE_INST_VALU / _VALU2)
spp=$1; shift; R=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03/ab
for v in "$@"; do
  if [ "$v" = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  echo "=== $v"
  python3 scripts/quick_bench.py $spp $R 2>&1 | grep -v "^\[rt stats\] waves\|timeline"
  D=gpurun_out/r03/ab/$v
  rm -rf $D; mkdir -p $D
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE --output-format csv -d $D -- python3 bench.py --pmc-child $( [ "$R" = mega ] && echo megakernel || echo wavefront ) --spp $spp > $D.out 2> $D.err
  python3 - <<PY
import csv, glob, collections
s = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$D/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s[r["Kernel_Name"].split("(")[0].replace("void rt::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in s.items():
    if c.get("SQ_INSTS_VALU", 0) < 1e8: continue
    n, cyc, a, a2 = c["SQ_INSTS_VALU"], c["GRBM_GUI_ACTIVE"] / 8, c["SQ_ACTIVE_INST_VALU"], c["SQ_ACTIVE_INST_VALU2"]
    print("   %s: VALU instr %.4g, cycles/instr/SIMD %.3f, paired %.3f, busy %.3f, kernel Mcycles %.1f" % (k, n, cyc * 1024 / n, a2 / n, 4 * (a - a2) / 1024 / cyc, cyc / 1e6))
PY
done
