#!/bin/bash
# k_wf_finish under rocprofv3 --pmc: one launch per sample (C=1) against the streaming schedule (C=8), 4K frame, 8 spp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/rqpmc && export TMPDIR=/tmp
sets=("SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
export RES=3840x2160
for c in 1 8; do
  export RT_WF_SAMPLES_PER_LAUNCH=$c
  for i in "${!sets[@]}"; do
    d=gpurun_out/rqpmc/c${c}_$i
    timeout -k 5 200 rocprofv3 --pmc ${sets[$i]} --kernel-trace -d $d -o out --output-format csv -- python3 scripts/quick_bench.py 8 wave > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, collections
for c in (1, 8):
    acc = collections.defaultdict(float); n = 0
    for f in glob.glob(f'gpurun_out/rqpmc/c{c}_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_wf_finish' in r['Kernel_Name']: acc[r['Counter_Name']] += float(r['Counter_Value'])
    print('C =', c, {k: f'{v:.4g}' for k, v in sorted(acc.items())})
PY
