"""Summarises a rocprofv3 kernel trace: per-kernel totals, wall span, and how much of the span had 0/1/2+ kernels running."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rt::" in r["Kernel_Name"]]
ev = []
tot = {}
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    tot[k] = tot.get(k, 0) + (b - a)
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
span = ev[-1][0] - ev[0][0]
hist = {}
cur = 0; last = ev[0][0]
for t, d in ev:
    hist[min(cur, 3)] = hist.get(min(cur, 3), 0) + (t - last)
    cur += d; last = t
print("span %.2f ms" % (span / 1e6))
for k, v in sorted(tot.items(), key=lambda x: -x[1]): print("  %-28s %.2f ms" % (k, v / 1e6))
for c in sorted(hist): print("  %d kernels running: %.2f ms (%.1f%%)" % (c, hist[c] / 1e6, 100 * hist[c] / span))
