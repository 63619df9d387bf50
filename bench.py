#!/usr/bin/env python3
"""bench.py — the headline benchmark: Mrays/s and ms/frame on the Sponza-like atrium, 1920x1080, 64 spp,
max depth 10 (BASELINE.json configs[3]), on N GPUs of one node.

A "step" is one full frame. One process per GPU (torch.distributed / RCCL): rank r renders the 8-row strips
k with k % N == r of the SAME frame (strong scaling: total work is fixed), then the per-rank strip buffers
are gathered to rank 0 over xGMI and de-interleaved there. Scene, BVH and ray queues are resident in HBM
before the timed region; the frame stays on the device (no PCIe traffic inside the timed region).

Both renderers of the path (wavefront, megakernel) are measured with the same K steps; the headline `value` is the
faster one (`config.renderer`), `renderers` holds both with their rooflines.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — dominant kernel of the headline renderer: algorithmic bytes / measured launch time vs the 8 TB/s HBM peak (the
                 contractual figure: tiny by construction for a traversal kernel), `traffic` = FETCH_SIZE + WRITE_SIZE of that
                 kernel, and `binding` = the bound that actually binds: VALU issue cycles against the ceiling measured for the
                 kernel's own instruction mix (profiles/r02_valu_calibration.json). traffic and binding are MEASURED IN THIS RUN
                 (N = 1): after the timed region bench.py renders one more frame of the same configuration in a child process
                 under `rocprofv3 --pmc` (three passes: FETCH_SIZE+TCC_HIT, WRITE_SIZE+TCC_MISS, SQ VALU counters); if rocprofv3
                 is not usable they fall back to the committed profiles/ files and say so in `source`.
  cpu_baseline — the CPU oracle (a port, not the reference: the reference cannot be built here) timed on the
                 host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Algorithmic bytes per traced ray. SURVEY.md §8(d) prices the reference's single shoot kernel at
# 76 B/ray (34 B state read + 8 B RNG r/w + 34 B state write). Here that kernel is split in two:
#   k_wf_extend (dominant, ~78 % of the frame): reads org+id 16 B + dir 8 B, writes the hit record 16 B  = 40 B/ray
#   k_wf_shade                                : reads state 34 B + hit 16 B + RNG 4 B, writes RNG 4 B + state 34 B = 92 B/ray
EXTEND_BYTES_PER_RAY = 40
SHADE_BYTES_PER_RAY = 92
# The wavefront renderer's default schedule (RT_WF_FINISH_DEPTH=0): one compacted queue per sample, written by
# k_wf_generate (34 B state + 4 B RNG r/w + count) and consumed by k_wf_finish, which follows every path to its end in registers:
#   k_wf_finish per PATH: reads ray state 34 B + RNG 4 B, writes RNG 4 B, reads + writes the accumulator 32 B       = 74 B/path
FINISH_BYTES_PER_PATH = 74
MEGAKERNEL_BYTES_PER_PIXEL = 16  # one fp32 RGBA store per pixel per frame (the timed region stores no RGBA8 image)
# SURVEY §8d "traversal bytes/ray" of this build: 64-B BVH4 nodes, 48-B triangle records, 124 B of hit shading;
# the per-ray counts are the atrium's, measured with RT_KERNEL_STATS=1 (gpurun_out/r02/stats_base.log): 17.10 nodes, 4.60 triangles
TRAVERSAL_BYTES_PER_RAY = 17.10 * 64 + 4.60 * 48 + 124


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--renderer", choices=["auto", "wavefront", "megakernel"], default=os.environ.get("RT_BENCH_RENDERER", "auto"),
                    help="auto = measure both renderers of the path, headline the faster one")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--detail", type=int, default=4, help="atrium tessellation (4 = ~281k triangles)")
    ap.add_argument("--bvh", choices=["sah", "lbvh"], default="sah")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="frames enqueued at once, each in its own renderer and strip buffer (1 = the reference's one frame at a "
                         "time; 2 hides the end-of-frame drain of the per-pixel sample chains behind the next frame)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the informational two-frames-in-flight measurement")
    ap.add_argument("--cpu-spp", type=int, default=8, help="samples per pixel of the bounded CPU-oracle sample")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes after the timed region")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # one frame, no timing: the process rocprofv3 wraps
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    from rtamd import abi, scenes
    from rtamd.renderer import Camera, MegakernelRenderer, Scene, WavefrontRenderer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL process group, gather, all-reduces) even with one
    # rank — a one-GPU rehearsal of what the N > 1 runs execute
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    W, H, strip = args.width, args.height, 8
    sd = scenes.atrium_scene(args.detail)
    scene = Scene(sd, device=local_rank, bvh=abi.RT_BVH_SAH if args.bvh == "sah" else abi.RT_BVH_LBVH)
    cam = Camera.for_scene(sd, (W, H))
    from rtamd.dist import FrameGatherer
    F0 = max(1, args.frames_in_flight)
    fgs = [FrameGatherer(rank, world, W, H, strip, dev, collective=use_dist) for _ in range(max(F0, 2))]  # one strip buffer per frame in flight
    fg = fgs[0]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(name, F=F0):
        """W warm-up frames, then exactly K timed frames of one renderer: render this rank's strips, RCCL gather,
        de-interleave on rank 0; barrier + synchronize on both sides; MAX over ranks of the time, SUM of the rays."""
        cls = WavefrontRenderer if name == "wavefront" else MegakernelRenderer
        rs = [cls(scene, (W, H), args.depth, args.spp) for _ in range(F)]
        for r in rs:
            r.set_tile(rank, world, strip)
            r.set_profiling(True)  # per-launch hipEvents of the dominant kernel, on the launch's own stream
        assert fg.rows == rs[0].local_rows
        tot = {"rays": 0, "hot_ms": 0.0, "hot_launches": 0, "dev_ms": 0.0}

        def collect(k):
            """wait for frame buffer k's frame, gather it (RCCL gather of the per-GPU strip buffers + de-interleave on rank 0)"""
            fr = rs[k].end_frame()
            fgs[k].gather()
            tot["rays"] += fr.rays
            tot["hot_ms"] += fr.hot_kernel_ms
            tot["hot_launches"] += fr.hot_kernel_launches
            tot["dev_ms"] += fr.device_ms

        def run(n):
            """n frames; with --frames-in-flight F > 1 frame i+1..i+F-1 are already enqueued (each in its own renderer and
            strip buffer) while frame i finishes and is gathered: the next frame's waves fill the end-of-frame drain"""
            queue = []
            for i in range(n):
                k = i % F
                if len(queue) == F:
                    collect(queue.pop(0))
                rs[k].begin_frame(cam, d_f32=fgs[k].local.data_ptr())
                queue.append(k)
            while queue:
                collect(queue.pop(0))

        run(args.warmup)
        barrier()
        for key in tot:
            tot[key] = 0
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        rays, hot_ms, hot_launches, dev_ms = tot["rays"], tot["hot_ms"], tot["hot_launches"], tot["dev_ms"]
        rows = rs[0].local_rows
        for r in rs:
            r.close()
        rays_all = float(rays)
        if use_dist:
            mx = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = torch.tensor([float(rays)], dtype=torch.float64, device=dev)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            elapsed, rays_all = float(mx[0]), float(sm[0])
        # roofline of the dominant kernel on THIS rank: algorithmic bytes per launch / mean launch time
        finish_depth = int(os.environ.get("RT_WF_FINISH_DEPTH", "0"))
        if name == "wavefront" and finish_depth < args.depth:  # the profiled launches are k_wf_finish: one per sample and stream lane
            kernel, bytes_per_launch = "k_wf_finish", FINISH_BYTES_PER_PATH * (rows * W * args.spp * args.steps) / max(hot_launches, 1)
        elif name == "wavefront":
            kernel, bytes_per_launch = "k_wf_extend", EXTEND_BYTES_PER_RAY * rays / max(hot_launches, 1)
        else:
            kernel, bytes_per_launch = "k_megakernel", MEGAKERNEL_BYTES_PER_PIXEL * rows * W
        mean_launch_s = (hot_ms / max(hot_launches, 1)) * 1e-3
        achieved = bytes_per_launch / mean_launch_s / 1e9 if mean_launch_s > 0 else 0.0
        roof = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "launches": hot_launches,
                "mean_launch_ms": round(mean_launch_s * 1e3, 5), "bytes_per_launch": round(bytes_per_launch, 1)}
        if name == "wavefront":
            if kernel == "k_wf_extend":
                frame_bytes = (EXTEND_BYTES_PER_RAY + SHADE_BYTES_PER_RAY) * rays_all / args.steps
                roof["frame_bytes_per_ray"] = EXTEND_BYTES_PER_RAY + SHADE_BYTES_PER_RAY
                roof["frame_achieved_gbs"] = round(frame_bytes / (elapsed / args.steps) / 1e9, 3)  # whole path, all kernels
            else:
                roof["bytes_per_path"] = FINISH_BYTES_PER_PATH
                per_lane = max(hot_launches, 1) / max(args.steps, 1)
                roof["schedule"] = (f"RT_WF_FINISH_DEPTH={finish_depth}: generate -> compacted queue -> k_wf_finish, {per_lane:g} launches per frame "
                                    f"for {args.spp} samples (the library picks samples per launch and stream lanes by tile size: DESIGN.md §5)")
            roof["concurrent_streams"] = int(os.environ.get("RT_WF_STREAMS", "2"))  # sub-tiles on streams of their own share the chip (1 in the chain regime)
            roof["note"] = "incoherent BVH traversal is bound by VALU issue at its lane utilisation; the HBM fraction is small by construction (DESIGN.md §6)"
        else:
            roof["traversal_bytes_per_ray"] = round(TRAVERSAL_BYTES_PER_RAY, 1)  # gathers, served by LDS / L1 / L2 / Infinity Cache
            roof["traversal_achieved_gbs"] = round(TRAVERSAL_BYTES_PER_RAY * rays_all / elapsed / 1e9, 1)
            roof["note"] = ("the megakernel has no per-ray HBM stream (SURVEY.md §8d): its only algorithmic HBM traffic is the frame store; "
                            "its BVH gathers are served by L2 / Infinity Cache (profiles/r01_traffic.json)")
        return {"renderer": name, "value": round(rays_all / elapsed / 1e6, 3), "unit": "Mrays/s",
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "rays_per_frame": int(rays_all / args.steps),
                "device_ms_per_step": round(dev_ms / args.steps, 3), "roofline": roof}

    names = ["wavefront", "megakernel"] if args.renderer == "auto" else [args.renderer]
    results = [measure(n) for n in names]
    # for information only, never the headline: the same renderer with two frames in flight
    lead = max(results, key=lambda d: d["value"])["renderer"] if rank == 0 else None
    if use_dist:
        box = [lead]
        dist.broadcast_object_list(box, src=0)
        lead = box[0]
    pipelined = measure(lead, 2) if F0 == 1 and not args.no_pipelined else None

    if rank == 0:
        best = max(results, key=lambda d: d["value"])  # headline: the faster of the path's two renderers
        out = {
            "metric": "Mrays/s", "value": best["value"], "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": best["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"atrium (Sponza-like, {sd.n_triangles} tris) {W}x{H} {args.spp}spp depth {args.depth}",
                       "renderer": best["renderer"], "bvh": args.bvh, "tile_split": f"{strip}-row strips, rank = strip % {world}", "frames_in_flight": F0,
                       "rays_per_frame": best["rays_per_frame"]},
            "roofline": best["roofline"],
            "device_ms_per_step": best["device_ms_per_step"],
            "renderers": {d["renderer"]: d for d in results},
        }
        if pipelined:
            out["two_frames_in_flight"] = {"renderer": pipelined["renderer"], "value": pipelined["value"], "unit": "Mrays/s",
                                           "ms_per_step": pipelined["ms_per_step"],
                                           "note": "same K frames with the next frame enqueued (second renderer + strip buffer) before the "
                                                   "current one is collected; not the headline: ms_per_step is then no longer one frame's latency"}
        if world == 1:
            attach_pmc(out["roofline"], best, args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, W, H, args.depth, args.cpu_spp, best["renderer"])
        print(json.dumps(out), flush=True)
    scene.close()
    if use_dist:
        dist.destroy_process_group()
    return 0


def pmc_child(args):
    """One untimed frame of the bench configuration (one renderer): the process `rocprofv3 --pmc ... --` wraps."""
    from rtamd import abi, scenes
    from rtamd.renderer import Camera, MegakernelRenderer, Scene, WavefrontRenderer
    sd = scenes.atrium_scene(args.detail)
    scene = Scene(sd, device=0, bvh=abi.RT_BVH_SAH if args.bvh == "sah" else abi.RT_BVH_LBVH)
    cam = Camera.for_scene(sd, (args.width, args.height))
    cls = WavefrontRenderer if args.renderer == "wavefront" else MegakernelRenderer
    r = cls(scene, (args.width, args.height), args.depth, args.spp)
    fr = r.render_frame(cam, want_f32=False, want_u8=False)
    print(json.dumps({"rays": fr.rays, "device_ms": fr.device_ms}), flush=True)
    r.close()
    scene.close()
    return 0


PMC_PASSES = (("FETCH_SIZE", "TCC_HIT_sum"), ("WRITE_SIZE", "TCC_MISS_sum"),
              ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"))


def live_pmc(args, renderer, kernel):
    """Counters of `kernel` over ONE frame of this very configuration, each pass in its own child process under rocprofv3 --pmc
    (counters only: no trace domains). Returns {counter: sum over the kernel's launches, "launches": n} or raises."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists():
        raise RuntimeError("rocprofv3 not found")
    child = [sys.executable, str(REPO / "bench.py"), "--pmc-child", "--renderer", renderer, "--width", str(args.width), "--height", str(args.height),
             "--spp", str(args.spp), "--depth", str(args.depth), "--detail", str(args.detail), "--bvh", args.bvh]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RT_BENCH_FORCE_DIST"):
        env.pop(k, None)
    sums, launches = {}, 0
    for counters in PMC_PASSES:
        with tempfile.TemporaryDirectory(prefix="rt_pmc_", dir="/tmp") as d:
            p = subprocess.run([exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=120)
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if p.returncode != 0 or not files:
                raise RuntimeError(f"rocprofv3 --pmc {' '.join(counters)} failed (rc {p.returncode}): {p.stderr[-300:]}")
            ids = set()
            for f in files:
                for row in csv.DictReader(open(f)):
                    if kernel in row["Kernel_Name"]:
                        sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                        ids.add(row["Dispatch_Id"])
            launches = max(launches, len(ids))
    sums["launches"] = launches
    return sums


def attach_pmc(roof, best, args):
    """roofline.traffic (HBM-side bytes per launch of the dominant kernel) and roofline.binding (the bound that binds)."""
    kernel, name = roof["kernel"], best["renderer"]
    # VALU issue ceiling of the kernel's DYNAMIC instruction mix: its three step kinds (inner node / whole leaf / shading round) priced
    # opcode by opcode with the calibrated issue costs and weighted with the measured steps per ray (scripts/phase_cost.py ->
    # profiles/r02_phase_cost.json; k_wf_finish and k_wf_extend run the same steps). The static whole-kernel mix of
    # profiles/r02_valu_calibration.json over-weights the long, rarely executed shading code and is only the fallback.
    ceiling, ceiling_source = None, None
    phase_file, calib_file = REPO / "profiles" / "r02_phase_cost.json", REPO / "profiles" / "r02_valu_calibration.json"
    if phase_file.exists():
        ceiling = json.loads(phase_file.read_text()).get("per_ray", {}).get("dynamic_ceiling_cycles_per_instruction")
        ceiling_source = "profiles/r02_phase_cost.json (dynamic mix: steps per ray x calibrated cost of each step kind)"
    if ceiling is None and calib_file.exists():
        ceil_key = "k_wf_extend_static_mix" if kernel == "k_wf_extend" else "k_megakernel_static_mix"
        ceiling = json.loads(calib_file.read_text()).get("valu_ceiling_cycles_per_instruction", {}).get(ceil_key)
        ceiling_source = "profiles/r02_valu_calibration.json (static whole-kernel mix)"
    c, source = None, None
    if not args.no_live_pmc:
        try:
            c = live_pmc(args, name, kernel)
            source = "measured in this run: one more frame of the same configuration in a child process under rocprofv3 --pmc (3 passes)"
        except Exception as e:  # no rocprofv3 / no counter access here: say so and use the committed profile of the same command
            source = f"live rocprofv3 --pmc unavailable ({str(e)[:160]}); "
    if c is None:
        prof = REPO / "profiles" / "r02_pmc_bench.json"
        if prof.exists():
            c = json.loads(prof.read_text()).get(kernel)
            source = (source or "") + "committed profiles/r02_pmc_bench.json (same command, scripts/final_profile.sh)"
    if not c:
        roof["traffic_note"] = (source or "") + "no counters available"
        return
    n = max(int(c.get("launches", 1)), 1)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0  # rocprofv3 reports both in KiB
        roof["traffic"] = round((fetch + write) / n, 1)
        roof["traffic_detail"] = {"fetch_bytes_per_launch": round(fetch / n, 1), "write_bytes_per_launch": round(write / n, 1),
                                  "fetch_x2_upper_bound_per_launch": round((2 * fetch + write) / n, 1),
                                  "bytes_per_ray": round((fetch + write) / max(best["rays_per_frame"], 1), 1),
                                  "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4) if c.get("TCC_HIT_sum") and c.get("TCC_MISS_sum") else None}
        roof["traffic_note"] = ("FETCH_SIZE + WRITE_SIZE of the kernel, separate --pmc passes, KiB -> bytes. MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts a wide "
                                "coalesced 16 B/lane stream at half its bytes; these reads are scattered 64-byte node / 48-byte triangle gathers (uncalibrated width), so the "
                                "raw figure is `traffic` and the x2 bound is given beside it. Infinity-Cache hits are included (the scene is MALL-resident): this is fabric "
                                "traffic, far above the algorithmic bytes because a traversal kernel's real work is gathers, not its output stream.")
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c and ceiling:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
        per_simd = c["SQ_INSTS_VALU"] / 1024.0       # 256 CUs x 4 SIMDs
        achieved_cpi = cycles / max(per_simd, 1.0)   # kernel cycles per VALU instruction a SIMD issued
        roof["binding"] = {"bound": "valu_issue", "frac": round(ceiling / achieved_cpi, 4),
                           "achieved_cycles_per_valu_instruction": round(achieved_cpi, 3), "ceiling_cycles_per_valu_instruction": ceiling,
                           "ceiling_source": ceiling_source,
                           "lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4) if c.get("SQ_ACTIVE_INST_VALU") else None,
                           "valu_instructions_per_ray": round(c["SQ_INSTS_VALU"] / max(best["rays_per_frame"], 1), 1),
                           "note": "frac = VALU issue cycles the kernel needs at the measured cost of its own instruction mix (profiles/r02_valu_calibration.json: a "
                                   "wave64 VALU instruction costs 2, 4 or 8 cycles by opcode on gfx950) / kernel cycles. ~1 = the SIMDs issue VALU instructions "
                                   "back to back: the kernel is bound by instruction issue at its lane utilisation, not by HBM, cache bandwidth or latency. The mix model is good to a "
                                   "few percent (it prices every conditional push of an inner step), so a value slightly above 1 reads as 1; nothing is clamped."}
    roof["pmc_source"] = source


def cpu_baseline(sd, W, H, depth, spp, renderer):
    """The CPU oracle (oracle/oracle_rt.cpp: a port of the reference's algorithm, std::thread over rows) on the
    same scene/camera/seeds at the same resolution but `spp` samples instead of 64 (rays/s does not depend on
    spp: benchmark_avg.csv:12,2,16,18 of the reference)."""
    from oracle import oracle as O
    from rtamd import abi
    osc = O.OracleScene(sd)
    cam = O.camera(W, H, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    kind = abi.RT_RENDERER_WAVEFRONT if renderer == "wavefront" else abi.RT_RENDERER_MEGAKERNEL
    cores = O.hardware_threads()
    t0 = time.perf_counter()
    _, _, rays = osc.render(cam, kind, depth, spp, use_bvh=True, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"same scene and camera, {W}x{H}, {spp} spp (of 64), depth {depth}: {rays} rays in {dt:.1f} s",
            "note": "CPU restatement of the reference (oracle/), not the reference's SYCL build"}


if __name__ == "__main__":
    sys.exit(main())
