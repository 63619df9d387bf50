"""Copies the round's evidence from gpurun_out/r02/final (scripts/final_profile.sh) into profiles/:
   r02_bench.json, r02_bench_under_rocprof.json, r02_bench_kernel_stats.csv, r02_pmc_bench.json, r02_valu_calibration.json"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
F, P = REPO / "gpurun_out" / "r02" / "final", REPO / "profiles"
shutil.copy(F / "bench.json", P / "r02_bench.json")
shutil.copy(F / "bench_under_rocprof.json", P / "r02_bench_under_rocprof.json")
for name in ("config_table.txt", "tile_scaling.txt"):
    if (F / name).exists():
        shutil.copy(F / name, P / ("r02_" + name))
stats = sorted(glob.glob(str(F / "kt" / "**" / "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)  # gpurun_out keeps earlier runs: newest
if stats:
    rows = list(csv.reader(open(stats[-1])))
    keep = [rows[0]] + [r for r in rows[1:] if "rt::" in r[0]]
    csv.writer(open(P / "r02_bench_kernel_stats.csv", "w", newline="")).writerows(keep)
pmc = {"source": "scripts/final_profile.sh: rocprofv3 --pmc passes (counters only) of `python3 bench.py --pmc-child --renderer R`, ONE frame of the bench configuration "
                 "(atrium 1920x1080, 64 spp, depth 10) per pass; counters summed over the kernel's launches; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them"}
for rend in ("megakernel", "wavefront"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    ids = collections.defaultdict(set)
    for d in glob.glob(str(F / f"pmc_{rend}_*")):
        if not Path(d).is_dir():
            continue
        files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)  # newest run
        for row in csv.DictReader(open(files[-1])) if files else []:
            m = re.search(r"rt::(k_\w+)", row["Kernel_Name"])
            if m:
                acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
                ids[m.group(1)].add((d, row["Dispatch_Id"]))
    for k, c in acc.items():
        n_pass = max(1, len({d for d, _ in ids[k]}))
        e = dict(c)
        e["launches"] = len(ids[k]) // n_pass
        pmc[k] = e
(P / "r02_pmc_bench.json").write_text(json.dumps(pmc, indent=1))
out = subprocess.run([sys.executable, str(REPO / "scripts" / "valu_calib_summary.py")], capture_output=True, text=True, check=True).stdout
(P / "r02_valu_calibration.json").write_text(out)
stats_log = F / "stats.log"  # RT_KERNEL_STATS=1 frame of the final kernels (scripts/final_profile.sh)
out = subprocess.run([sys.executable, str(REPO / "scripts" / "phase_cost.py")] + ([str(stats_log)] if stats_log.exists() else []), capture_output=True, text=True, check=True).stdout
(P / "r02_phase_cost.json").write_text(out)
print(open(P / "r02_bench.json").read()[:600])
