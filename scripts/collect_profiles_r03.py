"""Copies round 3's evidence from gpurun_out/r03/final (scripts/final_profile_r03.sh) into profiles/: r03_bench.json, r03_bench_under_rocprof.json,
r03_bench_kernel_stats.csv, r03_pmc_bench.json (the counters bench.py measured live, per kernel: its fallback when rocprofv3 cannot run), tables."""
import csv, glob, json, os, shutil
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
F, P = REPO / "gpurun_out" / "r03" / "final", REPO / "profiles"
shutil.copy(F / "bench.json", P / "r03_bench.json")
shutil.copy(F / "bench_under_rocprof.json", P / "r03_bench_under_rocprof.json")
if (F / "pmc_bench.json").exists():
    shutil.copy(F / "pmc_bench.json", P / "r03_pmc_bench.json")
for name in ("config_table.txt", "tile_scaling.txt", "schedules_64spp.txt"):
    if (F / name).exists():
        shutil.copy(F / name, P / ("r03_" + name))
if (F / "stats.log").exists():
    (P / "r03_wave_stats.txt").write_text("".join(l for l in open(F / "stats.log") if l.startswith("[rt stats]") or "Renderer" in l))
stats = sorted(glob.glob(str(F / "kt" / "**" / "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    rows = list(csv.reader(open(stats[-1])))
    keep = [rows[0]] + [r for r in rows[1:] if "rt::" in r[0]]
    csv.writer(open(P / "r03_bench_kernel_stats.csv", "w", newline="")).writerows(keep)
d = json.loads((F / "bench.json").read_text())
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup")}), {k: v["value"] for k, v in d["renderers"].items()})
