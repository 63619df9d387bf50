"""Copies the round's evidence from gpurun_out/r05/final (scripts/final_profile.sh) into profiles/r05_*."""
import csv, glob, json, os, shutil
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
F, P, R = REPO / "gpurun_out" / "r05" / "final", REPO / "profiles", "r05_"
for src, dst in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("pmc_bench.json", "pmc_bench.json"),
                 ("config_table.txt", "config_table.txt"), ("tile_scaling.txt", "tile_scaling.txt"), ("schedules_64spp.txt", "schedules_64spp.txt"),
                 ("tile_pmc.txt", "tile_pmc.txt"), ("stats.log", "wave_stats.txt"), ("pixel_slices.txt", "pixel_slices.txt"), ("tile_floor.txt", "tile_floor_final.txt"),
                 ("shoot_rate.txt", "shoot_rate.txt")):
    if (F / src).exists():
        text = (F / src).read_text()
        if src.endswith(".json") and src != "pmc_bench.json":  # the JSON line only
            text = "".join(l for l in text.splitlines(True) if l.startswith("{"))
        (P / (R + dst)).write_text(text)
stats = sorted(glob.glob(str(F / "kt" / "**" / "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    rows = list(csv.reader(open(stats[-1])))
    csv.writer(open(P / (R + "bench_kernel_stats.csv"), "w", newline="")).writerows([rows[0]] + [r for r in rows[1:] if "rt::" in r[0]])
d = json.loads((P / (R + "bench.json")).read_text())
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup")}), {k: v["value"] for k, v in d["renderers"].items()})
