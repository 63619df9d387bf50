"""BASELINE.json configs[1..4] on one GPU (4K frame of config[4]: this GPU's 1/8 tile): Mrays/s and ms/frame of both renderers.
   config_table.py sweep [ITERATIONS=6] [OUTDIR=gpurun_out/r05/sweep]: the reference's published sweep instead (benchmark.py:6-47: depth 10..50 at
   128 spp and 32..512 spp at depth 10, both renderers, 1920x1080; 6 runs per point, the first discarded) through the C++ `raytracer` CLI on the
   generated atrium and, for its minecraft.glb row, the voxel terrain; scraped with the reference's regular expressions (benchmark.py:49-55),
   CSV in the reference's column order, d = 10 -> 50 ratio beside the reference's (benchmark_avg.csv:2-11)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import scenes


def sweep(iters, out):
    import itertools, json, re, subprocess, tempfile, time
    from rtamd.glb_export import export_glb
    out.mkdir(parents=True, exist_ok=True)
    tmp = Path(tempfile.mkdtemp(prefix="rt_sweep_"))  # the exported scenes (tens of MB) and out.png stay out of the results directory
    exe = REPO / "sycl-ray-tracer_amd" / "host" / "build" / "raytracer"
    glbs = {"./assets/atrium4.glb": tmp / "atrium4.glb", "./assets/voxel4.glb": tmp / "voxel4.glb"}
    export_glb(scenes.atrium_scene(4), glbs["./assets/atrium4.glb"])
    export_glb(scenes.voxel_scene(4), glbs["./assets/voxel4.glb"])
    depth_samples = [(10, 128), (20, 128), (30, 128), (40, 128), (50, 128), (10, 32), (10, 128), (10, 256), (10, 512)]  # benchmark.py:6-18
    points = [("./assets/atrium4.glb", ds) for ds in depth_samples] + [("./assets/voxel4.glb", (10, 128))]
    hdr = "renderer,depth,samples,scene,time,rays_per_sec,ray_count\n"
    (out / "benchmark_raw.csv").write_text(hdr), (out / "benchmark_avg.csv").write_text(hdr)
    avg, t_all = {}, time.time()
    for (scene, (depth, samples)), renderer in itertools.product(points, ["-m", "-w"]):
        tot = [0.0, 0.0, 0]
        for i in range(iters):
            o = subprocess.check_output([str(exe), renderer, "-d", str(depth), "-s", str(samples), "--quiet", "--out", str(tmp / "out.png"), str(glbs[scene])], cwd=tmp).decode()
            if i == 0: continue  # benchmark.py:45-46
            rps = float(re.search(r"Rays/sec: (\d+\.\d+)M", o).group(1))
            secs = float(re.search(r"Time measured: (\d+\.\d+) seconds", o).group(1))
            rays = int(re.search(r"Total rays: (\d+)", o).group(1))
            tot[0] += secs; tot[1] += rps; tot[2] += rays
            with open(out / "benchmark_raw.csv", "a") as f: f.write(f"{renderer},{depth},{samples},{scene},{secs},{rps},{rays}\n")
        n = max(iters - 1, 1)
        with open(out / "benchmark_avg.csv", "a") as f: f.write(f"{renderer},{depth},{samples},{scene},{tot[0] / n},{tot[1] / n},{tot[2] / n}\n")
        avg[(renderer, depth, samples) if "atrium" in scene else (renderer, "voxel")] = (tot[0] / n, tot[1] / n, tot[2] / n)
        print(f"{scene} {renderer} d={depth} s={samples}: {tot[0] / n:.4f} s  {tot[1] / n:.1f} Mrays/s  {tot[2] / n:.0f} rays", flush=True)
    ref = {"-m": (694.284, 507.954), "-w": (461.804, 386.6)}  # benchmark_avg.csv:2-3,10-11 (Sponza, unstated Intel GPU)
    ref2 = {"-m": [317.8, 694.3], "-w": [161.2, 461.8]}      # benchmark_avg.csv:20-21 against :2-3
    summary = {"iterations_kept": iters - 1, "wall_s": round(time.time() - t_all, 1), "scenes": list(glbs), "depth_ratio_d50_over_d10": {}}
    for r in ("-m", "-w"):
        summary["depth_ratio_d50_over_d10"][r] = {"this_build": round(avg[(r, 50, 128)][1] / avg[(r, 10, 128)][1], 3), "reference": round(ref[r][1] / ref[r][0], 3),
                                                  "this_build_mrays_d10_d50": [round(avg[(r, 10, 128)][1], 1), round(avg[(r, 50, 128)][1], 1)], "reference_mrays_d10_d50": list(ref[r])}
    summary["spp_flatness_mrays_s32_s512"] = {r: [round(avg[(r, 10, s)][1], 1) for s in (32, 128, 256, 512)] for r in ("-m", "-w")}
    summary["second_scene_d10_s128"] = {r: {"this_build_mrays_voxel_vs_atrium": [round(avg[(r, "voxel")][1], 1), round(avg[(r, 10, 128)][1], 1)],
                                            "ratio": round(avg[(r, "voxel")][1] / avg[(r, 10, 128)][1], 3), "reference_mrays_minecraft_vs_sponza": ref2[r],
                                            "reference_ratio": round(ref2[r][0] / ref2[r][1], 3), "rays_per_path": round(avg[(r, "voxel")][2] / (1920 * 1080 * 128), 2)} for r in ("-m", "-w")}
    (out / "sweep_summary.json").write_text(json.dumps(summary, indent=1))
    print(json.dumps(summary, indent=1))


if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 6, (Path(sys.argv[3]) if len(sys.argv) > 3 else REPO / "gpurun_out" / "r05" / "sweep").resolve())
    sys.exit(0)
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
CASES = [("cube 512x512 16spp d10", scenes.cube_scene, {}, 512, 512, 16, 10, 1),
         ("cornell 1024x1024 64spp d10", scenes.cornell_scene, {}, 1024, 1024, 64, 10, 1),
         ("atrium 1920x1080 64spp d10", scenes.atrium_scene, {"detail": 4}, 1920, 1080, 64, 10, 1),
         ("atrium 3840x2160 256spp d16, tile 1/8", scenes.atrium_scene, {"detail": 4}, 3840, 2160, 256, 16, 8),
         ("voxel terrain 1920x1080 128spp d10", scenes.voxel_scene, {"detail": 4}, 1920, 1080, 128, 10, 1)]  # the reference's minecraft.glb row (benchmark_avg.csv:20-21)
for label, fn, kw, w, h, spp, depth, world in CASES:
    sd = fn(**kw); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (w, h))
    for cls in (WavefrontRenderer, MegakernelRenderer):
        r = cls(sc, (w, h), depth, spp)
        if world > 1: r.set_tile(0, world, 8)
        r.render_frame(cam, want_f32=False, want_u8=False)
        fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(2)), key=lambda f: f.device_ms)
        print(f"{label:40s} {cls.__name__:18s} rays {fr.rays:>12d}  {fr.device_ms:9.2f} ms  {fr.rays/fr.device_ms/1e3:8.1f} Mrays/s", flush=True)
        r.close()
    sc.close()
