#!/bin/bash
# VALU instruction count / issue counters of the megakernel and k_wf_finish for library variants (one 8 spp frame each):
#   scripts/variant_pmc.sh main pp ...
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/vpmc && export TMPDIR=/tmp
sets=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU")
for v in "$@"; do
  if [ $v = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  for i in "${!sets[@]}"; do
    d=gpurun_out/vpmc/${v}_$i
    timeout -k 5 120 rocprofv3 --pmc ${sets[$i]} --kernel-trace -d $d -o out --output-format csv -- python3 scripts/quick_bench.py 8 both > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  done
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys
for v in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for f in glob.glob(f'gpurun_out/vpmc/{v}_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = 'mega' if 'k_megakernel' in r['Kernel_Name'] else 'finish' if 'k_wf_finish' in r['Kernel_Name'] else None
            if k: acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for k in acc:
        a = {c: acc[k][c] / n[k][c] for c in acc[k]}
        print(v, k, {c: f'{x:.4g}' for c, x in sorted(a.items())}, 'cycles/VALU instr per SIMD %.3f' % (a['GRBM_GUI_ACTIVE'] / 8 / (a['SQ_INSTS_VALU'] / 1024)),
              'lane util %.3f' % (a['SQ_THREAD_CYCLES_VALU'] / 64 / a['SQ_ACTIVE_INST_VALU']))
PY
