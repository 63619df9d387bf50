#!/bin/bash
# counters of k_megakernel_ts against k_megakernel (one 8 spp frame each). usage: ts_pmc.sh [sq|mem]
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ts_pmc && export TMPDIR=/tmp
if [ "${1:-sq}" = mem ]; then
  sets=("TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum")
else
  sets=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS")
fi
for mode in ${MODES:-0 1}; do
  export RT_MEGA_TS=$mode
  for i in "${!sets[@]}"; do
    d=gpurun_out/ts_pmc/m${mode}_${1:-sq}$i
    timeout -k 5 120 rocprofv3 --pmc ${sets[$i]} --kernel-trace -d $d -o out --output-format csv -- python3 scripts/quick_bench.py 8 mega > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, collections
for mode in (0, 1):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f'gpurun_out/ts_pmc/m{mode}_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'megakernel' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print('mode', mode, {k: f'{v / n[k]:.4g}' for k, v in sorted(acc.items())}, 'launches', set(n.values()))
PY
