"""bench_summary.py FILE: the bench line's headline, rooflines, sensitivity and baselines in a few lines (for reading a GPU call's result)."""
import json
import sys

d = json.load(open(sys.argv[1]))
print("headline", d["value"], d["unit"], d["ms_per_step"], "ms", d["config"]["renderer"], "|", d["config"]["workload"])
for k, v in d["renderers"].items():
    r = v["roofline"]
    print(f"  {k:28s} {v['value']:9.1f} Mrays/s {v['ms_per_step']:8.2f} ms  roofline achieved {r['achieved']} GB/s frac {r['frac']}  per launch {r['per_launch']['achieved']} GB/s x{r['per_launch']['launches']}")
r = d["roofline"]
print("roofline keys", sorted(r))
for k in ("traffic", "traffic_detail", "traversal", "pixel_slices"):
    if k in r:
        print(" ", k, json.dumps(r[k])[:600])
if "binding" in r:
    b = r["binding"]
    print("  binding valu", b["frac"], "lanes", b.get("lane_utilisation"), "instr/ray", b.get("valu_instructions_per_ray"), "L1", json.dumps(b.get("vector_l1")))
print("two frames", json.dumps(d.get("two_frames_in_flight"))[:300])
print("cpu", json.dumps(d.get("cpu_baseline"))[:400])
ws = d.get("workload_sensitivity", {})
for k, v in ws.get("scenes", {}).items():
    print("  sens", k, json.dumps(v))
if "error" in ws:
    print("  sens error", ws["error"])
