#!/bin/bash
# kernel trace of the schedule probe: per-kernel launch counts, mean durations and the gaps between consecutive launches on a stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/ktrace_$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 scripts/stats_probe.py > $O/out.txt 2> $O/err.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"].split("(")[0].replace("void rt::", "")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in per.items():
    d = [e - s for s, e in v]
    print(f"{k[:60]:60s} n {len(v):5d} mean {sum(d)/len(d)/1e3:9.1f} us  total {sum(d)/1e6:8.2f} ms")
# gaps between consecutive launches of the SHOOT kernel in the measured frame
sh = [v for k, v in per.items() if "true>" in k and "k_wf_finish" in k]
if sh:
    v = sh[0][-40:]
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print("SHOOT last 40 launches: durations us", [round((e - s) / 1e3) for s, e in v])
    print("gaps us", [round(g / 1e3) for g in gaps])
PY
