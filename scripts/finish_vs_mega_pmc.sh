#!/bin/bash
# k_wf_finish as "the megakernel's loop fed from a queue" (one stream lane, all samples in one launch, pixel kept in its lane) against
# k_megakernel: same schedule, which counters differ? 1080p, 8 spp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/fvm && export TMPDIR=/tmp
export RT_WF_STREAMS=1 RT_WF_REQUEUE=0 RT_WF_SAMPLES_PER_LAUNCH=8
sets=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_ANY")
for i in "${!sets[@]}"; do
  d=gpurun_out/fvm/p$i
  timeout -k 5 200 rocprofv3 --pmc ${sets[$i]} --kernel-trace -d $d -o out --output-format csv -- python3 scripts/quick_bench.py 8 both > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob('gpurun_out/fvm/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = 'mega' if 'k_megakernel' in r['Kernel_Name'] else 'finish' if 'k_wf_finish' in r['Kernel_Name'] else None
        if k: acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for c in sorted(acc['mega']):
    m, f = acc['mega'][c] / n['mega'][c], acc['finish'][c] / n['finish'][c]
    print(f'{c:28s} mega {m:12.4g}  finish {f:12.4g}  ratio {f / m:.3f}')
PY
