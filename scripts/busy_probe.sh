#!/bin/bash
# VALU busy and dynamic instruction-type mix of the render kernels (round 3): rocprofv3 --pmc passes of one frame (bench.py --pmc-child).
#   usage (GPU box, repo root): bash scripts/busy_probe.sh <renderer> <spp> <tag> [extra bench args]
set -e
R=${1:-megakernel}; SPP=${2:-16}; TAG=${3:-probe}; shift 3 || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/busy_$TAG
mkdir -p $O
i=0
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
         "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/p$i -- python3 bench.py --pmc-child $R --spp $SPP "$@" > $O/p$i.out 2> $O/p$i.err || { echo "pass $i failed"; tail -3 $O/p$i.err; }
done
python3 - <<PY
import csv, glob, collections
s = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void rt::", "")
        s[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in s.items():
    if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
    n = c["SQ_INSTS_VALU"] / (2 if "SQ_INSTS_VALU_CVT" in c and "SQ_ACTIVE_INST_VALU" in c else 1)  # SQ_INSTS_VALU collected in two passes
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    a, a2 = c.get("SQ_ACTIVE_INST_VALU", 0), c.get("SQ_ACTIVE_INST_VALU2", 0)
    print(k, "| VALU instr %.4g | cycles/instr/SIMD %.3f | A/I %.3f A2/I %.3f | VALU busy 4(A-A2)/(1024 cyc) = %.3f" % (n, cyc * 1024 / n, a / n, a2 / n, 4 * (a - a2) / 1024 / cyc if cyc else 0))
    print("   mix: " + ", ".join("%s %.3f" % (t, c.get("SQ_INSTS_VALU_" + t, 0) / n) for t in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "CVT", "INT32")) + " | SALU/VALU %.3f" % (c.get("SQ_INSTS_SALU", 0) / n))
    print("   " + ", ".join("%s %.4g" % (t, c.get(t, 0)) for t in ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_SCA")))
PY
