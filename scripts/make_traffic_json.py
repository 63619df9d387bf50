"""profiles/r01_traffic.json from the counter CSVs of four rocprofv3 passes (scripts/pmc_pass.sh):
   python3 scripts/make_traffic_json.py gpurun_out/pmc_<tag>_{fetch,write,hit,miss}...  (any directories; all CSVs are merged)."""
import collections, csv, glob, json, os, re, sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
rays = None
for d in sys.argv[1:]:
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # gpurun merges runs into the same directory: only the newest pass counts
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            m = re.search(r"rt::(k_\w+)", k)
            if not m:
                continue
            acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[m.group(1)][row["Counter_Name"]].add(row["Dispatch_Id"])
    for f in glob.glob(d + ".log"):
        m = re.search(r'"rays_per_frame": (\d+)', open(f).read())
        if m:
            rays = int(m.group(1))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum / TCC_MISS_sum (separate passes) on `python3 bench.py --steps 1 --warmup 0 "
                 "--spp 4 --no-cpu-baseline --renderer R` (scripts/pmc_pass.sh, scripts/make_traffic_json.py), MI355X, round 1",
       "units": "FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB; bytes = value * 1024. gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
                "counts wide coalesced 16 B/lane streams at half their size; the reads here are mostly uncoalesced 16-byte BVH gathers (uncalibrated "
                "width), so both the raw figure and the x2 upper bound are given. Infinity-Cache hits are included in FETCH_SIZE (the 45 MB scene is "
                "MALL-resident), so this is fabric traffic, not DRAM traffic.",
       "rays_in_profile": rays}
for k in ("k_wf_extend", "k_wf_shade", "k_megakernel"):
    c = acc.get(k)
    if not c or "FETCH_SIZE" not in c:
        continue
    n = max(len(launches[k]["FETCH_SIZE"]), 1)
    fetch, write = c["FETCH_SIZE"] * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    e = {"launches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n, "traffic_bytes_per_launch": (fetch + write) / n,
         "traffic_bytes_per_launch_fetch_x2": (2 * fetch + write) / n}
    if rays:
        e["fetch_bytes_per_ray"], e["write_bytes_per_ray"] = fetch / rays, write / rays
    if c.get("TCC_HIT_sum") and c.get("TCC_MISS_sum"):
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if k == "k_megakernel":
        e["note"] = "spp 4 frame"
    out[k] = e
print(json.dumps(out, indent=1))
