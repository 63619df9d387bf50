"""The reference's published sweep (benchmark.py:6-47: depth 10..50 at 128 spp and spp 32..512 at depth 10, both renderers, 1920x1080) driven
through the C++ `raytracer` CLI on the generated atrium (the reference's Sponza asset is not shipped: .gitignore:4), scraped with the reference's
own regular expressions (benchmark.py:49-55) and written as CSV in the reference's column order (renderer,depth,samples,scene,time,rays_per_sec,ray_count).
    python3 scripts/sweep.py [ITERATIONS=6] [OUTDIR=gpurun_out/r03/sweep]      (GPU box; benchmark.py:35-68: 6 runs per point, the first discarded, mean of 5)
Both of the reference's scenes: its Sponza rows on the generated atrium, its minecraft.glb rows (benchmark.py:18: every (depth, samples) pair as well;
benchmark_avg.csv keeps d = 10, s = 128) on the generated voxel terrain.
Also prints the d = 10 -> 50 throughput ratio beside the reference's (benchmark_avg.csv:2-11: 0.73 megakernel, 0.84 wavefront)."""
import itertools, json, re, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import scenes
from rtamd.glb_export import export_glb

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
out = (Path(sys.argv[2]) if len(sys.argv) > 2 else REPO / "gpurun_out" / "r03" / "sweep").resolve()
out.mkdir(parents=True, exist_ok=True)
import tempfile
tmp = Path(tempfile.mkdtemp(prefix="rt_sweep_"))  # the exported scenes (tens of MB) and out.png stay out of the results directory
exe = REPO / "sycl-ray-tracer_amd" / "host" / "build" / "raytracer"
glbs = {"./assets/atrium4.glb": tmp / "atrium4.glb", "./assets/voxel4.glb": tmp / "voxel4.glb"}
if not glbs["./assets/atrium4.glb"].exists():
    export_glb(scenes.atrium_scene(4), glbs["./assets/atrium4.glb"])
if not glbs["./assets/voxel4.glb"].exists():
    export_glb(scenes.voxel_scene(4), glbs["./assets/voxel4.glb"])
depth_samples = [(10, 128), (20, 128), (30, 128), (40, 128), (50, 128), (10, 32), (10, 128), (10, 256), (10, 512)]  # benchmark.py:6-18
renderers = ["-m", "-w"]
# scene column: the generated Sponza stand-in (283,084 triangles) and Minecraft stand-in (260,366 triangles); no comma inside a CSV column
points = [("./assets/atrium4.glb", ds) for ds in depth_samples] + [("./assets/voxel4.glb", (10, 128))]
hdr = "renderer,depth,samples,scene,time,rays_per_sec,ray_count\n"
(out / "benchmark_raw.csv").write_text(hdr)
(out / "benchmark_avg.csv").write_text(hdr)
avg = {}
t_all = time.time()
for (scene, (depth, samples)), renderer in itertools.product(points, renderers):
    tot = [0.0, 0.0, 0]
    for i in range(iters):
        o = subprocess.check_output([str(exe), renderer, "-d", str(depth), "-s", str(samples), "--quiet", "--out", str(tmp / "out.png"), str(glbs[scene])], cwd=tmp).decode()
        if i == 0:
            continue  # benchmark.py:45-46
        rps = float(re.search(r"Rays/sec: (\d+\.\d+)M", o).group(1))
        secs = float(re.search(r"Time measured: (\d+\.\d+) seconds", o).group(1))
        rays = int(re.search(r"Total rays: (\d+)", o).group(1))
        tot[0] += secs; tot[1] += rps; tot[2] += rays
        with open(out / "benchmark_raw.csv", "a") as f:
            f.write(f"{renderer},{depth},{samples},{scene},{secs},{rps},{rays}\n")
    n = max(iters - 1, 1)
    with open(out / "benchmark_avg.csv", "a") as f:
        f.write(f"{renderer},{depth},{samples},{scene},{tot[0] / n},{tot[1] / n},{tot[2] / n}\n")
    if "atrium" in scene:
        avg[(renderer, depth, samples)] = (tot[0] / n, tot[1] / n, tot[2] / n)
    else:
        avg[(renderer, "voxel")] = (tot[0] / n, tot[1] / n, tot[2] / n)
    print(f"{scene} {renderer} d={depth} s={samples}: {tot[0] / n:.4f} s  {tot[1] / n:.1f} Mrays/s  {tot[2] / n:.0f} rays", flush=True)
ref = {"-m": (694.284, 507.954), "-w": (461.804, 386.6)}  # benchmark_avg.csv:2-3,10-11 (Sponza, unstated Intel GPU)
ref2 = {"-m": [317.8, 694.3], "-w": [161.2, 461.8]}      # benchmark_avg.csv:20-21 against :2-3
summary = {"iterations_kept": iters - 1, "wall_s": round(time.time() - t_all, 1), "scenes": list(glbs), "depth_ratio_d50_over_d10": {}}
for r in renderers:
    ours = avg[(r, 50, 128)][1] / avg[(r, 10, 128)][1]
    summary["depth_ratio_d50_over_d10"][r] = {"this_build": round(ours, 3), "reference": round(ref[r][1] / ref[r][0], 3),
                                              "this_build_mrays_d10_d50": [round(avg[(r, 10, 128)][1], 1), round(avg[(r, 50, 128)][1], 1)],
                                              "reference_mrays_d10_d50": list(ref[r])}
summary["spp_flatness_mrays_s32_s512"] = {r: [round(avg[(r, 10, s)][1], 1) for s in (32, 128, 256, 512)] for r in renderers}
# the reference's second scene: its rate on minecraft.glb relative to Sponza at d = 10, s = 128 (benchmark_avg.csv:2-3,20-21: 317.8 / 694.3, 161.2 / 461.8)
summary["second_scene_d10_s128"] = {r: {"this_build_mrays_voxel_vs_atrium": [round(avg[(r, "voxel")][1], 1), round(avg[(r, 10, 128)][1], 1)],
                                        "ratio": round(avg[(r, "voxel")][1] / avg[(r, 10, 128)][1], 3),
                                        "reference_mrays_minecraft_vs_sponza": ref2[r], "reference_ratio": round(ref2[r][0] / ref2[r][1], 3),
                                        "rays_per_path": round(avg[(r, "voxel")][2] / (1920 * 1080 * 128), 2)} for r in renderers}
(out / "sweep_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1))
