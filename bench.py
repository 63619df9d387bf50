#!/usr/bin/env python3
"""bench.py — the headline benchmark: Mrays/s and ms/frame on the Sponza-like atrium, 1920x1080, 64 spp,
max depth 10 (BASELINE.json configs[3]), on N GPUs of one node.

A "step" is one full frame. One process per GPU (torch.distributed / RCCL): rank r renders the 8-row strips
k with k % N == r of the SAME frame (strong scaling: total work is fixed), then the per-rank strip buffers
are gathered to rank 0 over xGMI and de-interleaved there. Scene, BVH and ray queues are resident in HBM
before the timed region; the frame stays on the device (no PCIe traffic inside the timed region).

`python3 bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment LAUNCHES ITS OWN RANKS: before anything touches the
GPU (this process never imports torch) it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py` as
a child process, relays its JSON line and exits with its code. Under an external launcher (WORLD_SIZE set) it is one rank.

Four renderers of the path are measured with the same K steps:
  wavefront                   the WavefrontRenderer's default schedule: one k_wf_generate + one k_wf_finish launch per frame
  megakernel                  the MegakernelRenderer
  wavefront_per_bounce        the WavefrontRenderer in the reference's shape (src/render_wavefront.cpp:396-417), a launch per sample and
                              bounce with the survivors compacted between bounces (__ballot / mbcnt ranks): k_wf_extend + k_wf_shade
  wavefront_per_bounce_fused  the same with ONE kernel per bounce (intersect + shade + compact: the reference's shoot_rays, :222-312)
The headline `value` is the fastest (`config.renderer`); `renderers` holds all four with their rooflines.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline     — of the headline renderer, as SURVEY §8(d) states it: algorithmic bytes per unit of work x units per second of the whole timed
                 region against the 8 TB/s HBM peak (tiny by construction for a traversal kernel), the dominant kernel's own per-launch
                 figure under `per_launch`; `traffic` = FETCH_SIZE + WRITE_SIZE of that kernel; `binding` = the bounds that actually bind,
                 MEASURED: VALU issue busy = 4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / (SIMDs x kernel cycles) and the vector L1's
                 tag look-ups per cycle and CU — counters of the product build in this very run, no model (semantics pinned on known
                 instruction streams: profiles/r03_valu_calibration2.json). traffic and binding are measured at N = 1: after the timed
                 region one more frame per renderer is rendered in a child process under `rocprofv3 --pmc`, one pass per counter group.
  cpu_baseline — the CPU oracle (a port, not the reference: the reference cannot be built here) timed on the
                 host cores on a bounded sample of the same workload (rank 0, N = 1 only)
  workload_sensitivity — information only (N = 1): the megakernel with the same K frames on the atrium turned out of the coordinate axes and on the
                 voxel terrain, each with node visits / triangle tests per ray: the bench scene is this traversal's best case.
For N > 1 the line also carries `per_rank_ms` (every rank's own time for the K frames), `gather_ms` (rank 0's mean time in the
gather + de-interleave per frame) and `rccl_ranks` (an all-reduce of ones over the process group: what RCCL saw).

Rehearsal without GPUs (tests/test_multi_gloo.py): `--backend gloo --rehearsal-tiles DIR` — the launcher, the process group, the
gather, the de-interleave and the JSON line run on the CPU with tiles read from DIR/tile_<rank>.npy instead of rendered ones.
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
# 76 B/ray (34 B state read + 8 B RNG r/w + 34 B state write). In the per-bounce schedule that kernel is split in two:
#   k_wf_extend (dominant in time): reads org+id 16 B + dir 8 B, writes the hit record 16 B                          = 40 B/ray
#   k_wf_shade                    : reads state 34 B + hit 16 B + RNG 4 B, writes RNG 4 B + state 34 B                = 92 B/ray
# and the fused per-bounce kernel (k_wf_finish limited to one bounce, "SHOOT") moves exactly SURVEY's 76 B/ray.
EXTEND_BYTES_PER_RAY = 40
SHADE_BYTES_PER_RAY = 92
SHOOT_BYTES_PER_RAY = 76
# The wavefront renderer's default schedule: one queue per frame, written by k_wf_generate and consumed by k_wf_finish, which
# keeps a pixel in its lane for all its samples (every camera ray but the first is made in the kernel) and follows every path to its end in
# registers. Its algorithmic stream, per PIXEL: the queue entry 34 B + the RNG word read and written back 8 B = 42 B; per PATH: the accumulator
# read and written, 32 B (src/render_wavefront.cpp:340-354 does the same read-modify-write in K5).
FINISH_BYTES_PER_PIXEL = 42
FINISH_ACCUM_BYTES_PER_PATH = 32
# the megakernel's HBM stream is its output — one fp32 RGBA store (16 B) and, as the reference's K1 writes it too
# (src/render_megakernel.cpp:154-158), one RGBA8 store (4 B) per pixel and frame — plus, since round 5, the pixel state that changes lanes
# between two slices of a pixel's samples (rt_kernels.h: SliceDev): 32 B written + 32 B read per hand-over.
MEGAKERNEL_BYTES_PER_PIXEL = 20
SLICE_HANDOVER_BYTES = 64
# k_wf_finish<.., SLICED> (the one-launch schedule on tiles with more pixels than lanes, round 5) keeps a pixel's sums in LDS within a slice:
# nothing per path; per PIXEL the 42 B above + the accumulator written once (16 B) + 64 B per hand-over
FINISH_SLICED_ACCUM_BYTES_PER_PIXEL = 16
# kernel names as rocprofv3 prints them, up to the template arguments that tell the instantiations apart (STATS, REQ, LIMIT; the fourth,
# SLICED, follows the frame's geometry): matched as substrings of the profiler's kernel names
FINISH_KERNEL, SHOOT_KERNEL = "k_wf_finish<false, false, false", "k_wf_finish<false, false, true"
# SURVEY §8d "traversal bytes/ray" of this build: 64-B BVH4 nodes, 40-B triangle records, 124 B of hit shading; per-ray counts of the
# atrium from the kernel's own counters (RT_KERNEL_STATS=1; re-measured in every run by the workload_sensitivity leg): 17.28 node visits,
# 9.28 of them in the LDS-staged top of the tree, 2.36 triangle tests
ATRIUM_VISITS = {"node_visits_per_ray": 17.28, "node_visits_in_lds_per_ray": 9.28, "triangle_tests_per_ray": 2.36}
NODE_BYTES, TRI_BYTES, HIT_SHADING_BYTES = 64, 40, 124
LDS_PEAK_GBS = 150000.0          # MI355X_MICROARCH.md (LDS): ~150 TB/s aggregate for ds_read_b64 / b128 with every CU streaming
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md (L2): ~34.5 TB/s aggregate
N_SIMDS = 1024                   # 256 CUs x 4
N_CUS = 256
RENDERERS = ("wavefront", "megakernel", "wavefront_per_bounce", "wavefront_per_bounce_fused")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--renderer", choices=["auto", *RENDERERS], default=os.environ.get("RT_BENCH_RENDERER", "auto"),
                    help="auto = measure all renderers of the path, headline the fastest")
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
    ap.add_argument("--pipelined-renderer", default="", help="the renderer of the two-frames-in-flight leg (default: the headline's)")
    ap.add_argument("--cpu-spp", type=int, default=8, help="samples per pixel of the bounded CPU-oracle sample")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes after the timed region")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the information-only workload_sensitivity leg (the megakernel on the rotated atrium and the voxel terrain)")
    ap.add_argument("--stats-child", default=None, help=argparse.SUPPRESS)  # comma-separated scenes: one 4-spp frame each under RT_KERNEL_STATS=1 (node visits / triangle tests per ray)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo: CPU rehearsal of the multi-rank path (needs --rehearsal-tiles)")
    ap.add_argument("--rehearsal-tiles", default=None, help="directory with tile_<rank>.npy (rows x W x 4 float32): used INSTEAD of rendering")
    ap.add_argument("--dump-frame", default=None, help="rank 0 writes the last gathered frame here (.npy)")
    ap.add_argument("--dump-pmc", default=None, help="write the raw counter sums of the live --pmc passes here (JSON; profiles/r03_pmc_bench.json is one)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)  # comma-separated renderers: one untimed frame each, the process rocprofv3 wraps
    return ap.parse_args(argv)


# ---- the launcher: N > 1 without an external one ----------------------------------------------------------------------------------
def launch_ranks(args) -> int:
    """Starts N ranks of this script with torch.distributed.run as a CHILD process (never an exec: nothing in this process has touched
    the GPU, and nothing will). Returns the child's exit code; its stdout (the JSON line of rank 0) is passed through."""
    # --standalone: the launcher picks and OWNS a free rendezvous port (binding port 0 here and handing the number on would leave a
    # window in which another process can take it: concurrent runs on one box collided that way)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env)
    return p.returncode


_JSON_FD = None


def emit(line: str) -> None:
    """The ONE line of the contract, on the process's real stdout."""
    os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())


def main() -> int:
    args = parse_args()
    # stdout carries the JSON line and nothing else: RCCL prints a version banner to stdout when its first communicator comes up, and
    # libraries may say more. File descriptor 1 is pointed at stderr for the life of the process; emit() writes to the saved one.
    global _JSON_FD
    launcher = "WORLD_SIZE" not in os.environ and args.gpus > 1  # (the launcher's child ranks do this themselves)
    if _JSON_FD is None and not args.pmc_child and not args.stats_child and not launcher:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)
    # HIP serves streams from GPU_MAX_HW_QUEUES hardware queues (4 by default): the per-bounce renderers' three lane streams, the frame's stream
    # and torch's own would share them and run one after the other (rt_frame.hip: wf_stream_lanes). Before torch or the library touch HIP:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.pmc_child:
        return pmc_child(args)
    if args.stats_child:
        return stats_child(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)  # before any torch / HIP call in this process

    import numpy as np
    import torch
    import torch.distributed as dist

    from rtamd.dist import FrameGatherer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rehearsal = args.backend == "gloo"
    if rehearsal and not args.rehearsal_tiles:
        raise SystemExit("--backend gloo is the CPU rehearsal of the multi-rank path: give --rehearsal-tiles DIR (there is no CPU renderer)")
    if rehearsal:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, gather, all-reduces) even with one rank
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    W, H, strip = args.width, args.height, 8
    F0 = max(1, args.frames_in_flight)
    # strip buffers: one per frame in flight + ONE more, so that the exchange of frame i (async gather, rtamd/dist.py) overlaps with
    # the rendering of frame i + 1 into another buffer
    fgs = [FrameGatherer(rank, world, W, H, strip, dev, collective=use_dist) for _ in range(max(F0, 2) + 1)]
    fg = fgs[0]

    def barrier():
        if use_dist:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    # what RCCL (gloo in the rehearsal) saw: every rank contributes a one
    comm_ranks = 1
    if use_dist:
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        comm_ranks = int(ones.item())

    setup = {"scene_s": 0.0, "bvh_build_upload_s": 0.0, "renderers_s": 0.0}  # this rank's time before the timed regions
    scene_digests = None
    if rehearsal:
        tile = torch.from_numpy(np.load(Path(args.rehearsal_tiles) / f"tile_{rank}.npy"))
        assert tile.shape[0] == fg.rows and tile.shape[1] == W, (tile.shape, fg.rows, W)
        scene = cam = None
        # the shared-scene path of the real run, rehearsed: generated by rank 0, read by the others, and every rank says what it holds
        t_setup = time.perf_counter()
        sd = shared_scene(args, rank, world, use_dist)
        setup["scene_s"] = round(time.perf_counter() - t_setup, 3)
        import hashlib
        mine = hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for a in (sd.positions, sd.normals, sd.uvs, sd.indices, sd.tri_instance, sd.textures))).hexdigest()
        scene_digests = [mine]
        if use_dist:
            scene_digests = [None] * world
            dist.all_gather_object(scene_digests, mine)
    else:
        from rtamd import abi, scenes
        from rtamd.renderer import Camera, MegakernelRenderer, Scene, WavefrontRenderer
        t_setup = time.perf_counter()
        sd = shared_scene(args, rank, world, use_dist)
        setup["scene_s"] = round(time.perf_counter() - t_setup, 3)
        t_setup = time.perf_counter()
        scene = Scene(sd, device=local_rank, bvh=abi.RT_BVH_SAH if args.bvh == "sah" else abi.RT_BVH_LBVH)
        setup["bvh_build_upload_s"] = round(time.perf_counter() - t_setup, 3)
        cam = Camera.for_scene(sd, (W, H))

    last_buffer = [0]  # the strip buffer of the last frame rendered (rank 0's gathered frame lives in that FrameGatherer)

    def make_renderer(name):
        cls = MegakernelRenderer if name == "megakernel" else WavefrontRenderer
        r = cls(scene, (W, H), args.depth, args.spp)
        if name.startswith("wavefront_per_bounce"):
            r.set_schedule(finish_depth=abi.RT_SCHED_ALL_BOUNCES, fused_bounce=name.endswith("fused"))
        r.set_tile(rank, world, strip)
        # per-launch hipEvents of the traversal / shading kernels, on the launches' own streams, INSIDE the timed region — for the
        # renderers with one such launch per frame. The per-bounce schedules have ~1,300 - 3,800 launches per frame: three event records
        # per launch pair on three streams slow them by a third (measured: 5.3 -> 3.1 Grays/s), so their timed frames run without events
        # and ONE more frame with events, after the timed region, supplies the per-kernel durations of their roofline objects.
        r.set_profiling(not name.startswith("wavefront_per_bounce"))
        return r

    def measure(name, F=F0):
        """W warm-up frames, then exactly K timed frames of one renderer: render this rank's strips, gather (RCCL), de-interleave on
        rank 0; barrier + synchronize on both sides; MAX over ranks of the time, SUM of the rays."""
        t_r = time.perf_counter()
        rs = [] if rehearsal else [make_renderer(name) for _ in range(F)]
        setup["renderers_s"] = round(setup["renderers_s"] + time.perf_counter() - t_r, 3)
        NB = len(fgs)
        if rs:
            assert fg.rows == rs[0].local_rows
        tot = {"rays": 0, "hot_ms": 0.0, "hot_launches": 0, "dev_ms": 0.0, "gather_s": 0.0, "kernel_ms": {}, "kernels": {}, "lanes": 0}

        def collect(k, b):
            """wait for renderer k's frame in strip buffer b, START its exchange (gather of the per-GPU strip buffers; the de-interleave on
            rank 0 follows in gather_end, when buffer b is needed again or the run ends)"""
            if rehearsal:
                fgs[b].local[: tile.shape[0]].copy_(tile)
                rays = 0
            else:
                fr = rs[k].end_frame()
                rays = fr.rays
                tot["hot_ms"] += fr.hot_kernel_ms
                tot["hot_launches"] += fr.hot_kernel_launches
                tot["dev_ms"] += fr.device_ms
                tot["lanes"] = fr.stream_lanes
                tot["pixel_slices"] = fr.pixel_slices
                for kn, v in fr.kernel_ms.items():
                    tot["kernel_ms"][kn] = tot["kernel_ms"].get(kn, 0.0) + v
                for kn, v in fr.kernels.items():
                    tot["kernels"][kn] = tot["kernels"].get(kn, 0) + v
            t = time.perf_counter()
            fgs[b].gather_begin()
            tot["gather_s"] += time.perf_counter() - t
            tot["rays"] += rays

        def run(n):
            """n frames; with --frames-in-flight F > 1 frame i+1..i+F-1 are already enqueued (each in its own renderer and
            strip buffer) while frame i finishes and is gathered: the next frame's waves fill the end-of-frame drain"""
            queue = []
            for i in range(n):
                k, b = i % F, i % NB
                if len(queue) == F:
                    collect(*queue.pop(0))
                t = time.perf_counter()
                fgs[b].gather_end()  # the exchange that last used this buffer (NB - 1 frames ago): long complete, this is where it is retired
                tot["gather_s"] += time.perf_counter() - t
                if not rehearsal:  # the frame as the reference's K1 writes it: fp32 tile AND the RGBA-unorm8 image (src/render_megakernel.cpp:154-158)
                    rs[k].begin_frame(cam, d_f32=fgs[b].local.data_ptr(), d_u8=fgs[b].local_u8.data_ptr())
                queue.append((k, b))
            while queue:
                collect(*queue.pop(0))
            t = time.perf_counter()
            for g in fgs:
                g.gather_end()
            tot["gather_s"] += time.perf_counter() - t
            if n:
                last_buffer[0] = (n - 1) % NB

        run(args.warmup)
        barrier()
        tot.update(rays=0, hot_ms=0.0, hot_launches=0, dev_ms=0.0, gather_s=0.0, kernel_ms={}, kernels={})
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        rows = fg.rows
        kernel_frames = args.steps
        if rs and name.startswith("wavefront_per_bounce"):  # the per-kernel durations: one profiled frame outside the timed region
            rs[0].set_profiling(True)
            fr = rs[0].render_frame_device(cam, d_f32=fgs[0].local.data_ptr(), d_u8=fgs[0].local_u8.data_ptr())
            tot["kernel_ms"], tot["hot_ms"], tot["hot_launches"], kernel_frames = dict(fr.kernel_ms), fr.hot_kernel_ms, fr.hot_kernel_launches, 1
            tot["profiled_rays"] = fr.rays
        for r in rs:
            r.close()
        rays_all, own_ms = float(tot["rays"]), elapsed * 1e3
        per_rank_ms = [round(own_ms, 3)]
        if use_dist:
            mx = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = torch.tensor([float(tot["rays"])], dtype=torch.float64, device=dev)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            each = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(each, torch.tensor([own_ms], dtype=torch.float64, device=dev))
            elapsed, rays_all, per_rank_ms = float(mx[0]), float(sm[0]), [round(float(e[0]), 3) for e in each]
        res = {"renderer": name, "value": round(rays_all / elapsed / 1e6, 3), "unit": "Mrays/s",
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "rays_per_frame": int(rays_all / args.steps),
               "device_ms_per_step": round(tot["dev_ms"] / args.steps, 3),
               "per_rank_ms": per_rank_ms, "gather_ms": round(tot["gather_s"] / args.steps * 1e3, 3)}
        if not rehearsal:
            res["launches_per_frame"] = {k: v // args.steps for k, v in tot["kernels"].items() if v}
            res["roofline"] = roofline_of(name, args, tot, rows, W, rays_all, elapsed, kernel_frames)
        return res

    if rehearsal:
        results = [measure("rehearsal")]
        lead, pipelined = "rehearsal", None
    else:
        names = list(RENDERERS) if args.renderer == "auto" else [args.renderer]
        results = [measure(n) for n in names]
        # for information only, never the headline: the same renderer with two frames in flight
        lead = max(results, key=lambda d: d["value"])["renderer"] if rank == 0 else None
        if use_dist:
            box = [lead]
            dist.broadcast_object_list(box, src=0)
            lead = box[0]
        pipelined = None
        if F0 == 1 and not args.no_pipelined:
            try:  # information only: never the reason a bench line is missing (with several ranks a failure here would have to be collective: let it raise)
                pipelined = measure(args.pipelined_renderer or lead, 2)
            except Exception as e:
                if use_dist:
                    raise
                pipelined = {"renderer": args.pipelined_renderer or lead, "value": None, "ms_per_step": None, "device_ms_per_step": None, "error": str(e)[:300]}

    setup_all = [setup]
    if use_dist:
        setup_all = [None] * world
        dist.all_gather_object(setup_all, setup)
    if rank == 0:
        best = max(results, key=lambda d: d["value"])  # headline: the fastest of the path's renderers
        workload = (f"atrium (Sponza-like, {sd.n_triangles} tris) {W}x{H} {args.spp}spp depth {args.depth}" if not rehearsal
                    else f"REHEARSAL (no rendering: tiles from files) {W}x{H}")
        out = {
            "metric": "Mrays/s", "value": best["value"], "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": best["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "renderer": best["renderer"], "bvh": args.bvh, "tile_split": f"{strip}-row strips, rank = strip % {world}",
                       "frames_in_flight": F0, "rays_per_frame": best["rays_per_frame"]},
            "roofline": best.get("roofline"),
            "device_ms_per_step": best["device_ms_per_step"],
            "per_rank_ms": best["per_rank_ms"], "gather_ms": best["gather_ms"],
            "rccl_ranks": comm_ranks, "backend": args.backend if use_dist else None,
            "setup_s_per_rank": setup_all, "scene_digests": scene_digests,  # before the timed regions: scene generated (rank 0) or read from rank 0's file, BVH built + uploaded, renderers created
            "renderers": {d["renderer"]: d for d in results},
        }
        if pipelined:
            out["two_frames_in_flight"] = {"renderer": pipelined["renderer"], "value": pipelined["value"], "unit": "Mrays/s",
                                           "ms_per_step": pipelined["ms_per_step"], "device_ms_per_step": pipelined["device_ms_per_step"],
                                           "note": "same K frames with the next frame enqueued (second renderer + strip buffer) before the "
                                                   "current one is collected; not the headline: ms_per_step is then no longer one frame's latency"}
            if pipelined.get("error"):
                out["two_frames_in_flight"]["error"] = pipelined["error"]
        if world == 1 and not rehearsal and not args.no_sensitivity and "megakernel" in out["renderers"]:
            try:
                out["workload_sensitivity"] = workload_sensitivity(args, out["renderers"]["megakernel"]["value"], abi, Scene, Camera, MegakernelRenderer)
                mine = out["workload_sensitivity"]["scenes"].get("atrium", {})
                if "node_visits_per_ray" in mine:  # this run's counters instead of the constants (the tree and the rays are the same for both renderers)
                    for mk in (out["renderers"].get(n) for n in ("megakernel", "wavefront")):
                        if mk:
                            mk["roofline"]["traversal"] = traversal_object({k: mine[k] for k in ATRIUM_VISITS}, mk["value"] * 1e6, "measured in this run (RT_KERNEL_STATS=1 child, 4 spp, megakernel)")
            except Exception as e:  # information only: never the reason a bench line is missing
                out["workload_sensitivity"] = {"error": str(e)[:300]}
        if world == 1 and not rehearsal:
            attach_pmc(out, args)
        if world == 1 and not rehearsal and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, W, H, args.depth, args.cpu_spp, best["renderer"])
        if args.dump_frame:
            np.save(args.dump_frame, fgs[last_buffer[0]].frame.cpu().numpy())
        sys.stdout.flush()
        emit(json.dumps(out))
    if scene is not None:
        scene.close()
    if use_dist:
        dist.destroy_process_group()
    return 0


def shared_scene(args, rank, world, use_dist):
    """The bench scene, GENERATED once per run: rank 0 writes it to one file, the other ranks read it (1.1 s of Python per rank
    otherwise, N times concurrently on one host). Every rank then builds its own BVH replica from it (rt_scene_create: 0.5 s)."""
    import torch.distributed as dist
    from rtamd import scenes
    if not (use_dist and world > 1):
        return scenes.atrium_scene(args.detail)
    box = [None]
    if rank == 0:
        fd, shared = tempfile.mkstemp(prefix="rt_bench_scene_", suffix=".npz")
        os.close(fd)
        sd = scenes.atrium_scene(args.detail)
        scenes.save_scene(sd, shared)
        box = [shared]
    dist.broadcast_object_list(box, src=0)  # (also the barrier: the file is complete when its name arrives)
    if rank != 0:
        sd = scenes.load_scene(box[0])
    dist.barrier()
    if rank == 0:
        os.unlink(box[0])
    return sd


def roofline_of(name, args, tot, rows, W, rays_all, elapsed, kernel_frames):
    """The roofline object of one renderer on THIS rank, as SURVEY §8(d) states it: algorithmic bytes per unit of work x units per second
    over the WHOLE timed region (wall clock: every launch, tail and gap included) against the HBM peak —
        wavefront renderers: rays/s x bytes per ray (76 B fused per-bounce kernel; 40 + 92 B EXTEND + SHADE; one-launch schedule: per PATH 32 B of accumulator + 42 B per pixel / spp, or, sliced, (58 B + 64 B per hand-over) per pixel / spp)
        megakernel         : pixels/s x (20 B + 64 B per slice hand-over)
    `per_launch` keeps the dominant kernel's own figure (its bytes / its mean launch duration from hipEvents on the launch's stream: what
    `rocprofv3 --kernel-trace --stats` reports for that kernel); where stream lanes overlap, launch durations overlap too, which is why that
    figure is not the frame's."""
    hot_launches, hot_ms = tot["hot_launches"], tot["hot_ms"]
    rays = tot.get("profiled_rays", tot["rays"])  # rays of the frames the kernel durations were taken over
    pixels = rows * W
    if name == "wavefront":
        slices = max(int(tot.get("pixel_slices", 1)), 1)
        per_path = round(FINISH_ACCUM_BYTES_PER_PATH + FINISH_BYTES_PER_PIXEL / args.spp if slices == 1 else
                         (FINISH_BYTES_PER_PIXEL + FINISH_SLICED_ACCUM_BYTES_PER_PIXEL + SLICE_HANDOVER_BYTES * (slices - 1)) / args.spp, 3)
        kernel, unit, units_per_s, bytes_per_unit = FINISH_KERNEL, "path", pixels * args.spp * args.steps / elapsed, per_path
        bytes_per_launch = per_path * (pixels * args.spp * args.steps) / max(hot_launches, 1)
    elif name == "wavefront_per_bounce":
        kernel, unit, units_per_s, bytes_per_unit = "k_wf_extend", "ray", rays_all / elapsed, EXTEND_BYTES_PER_RAY + SHADE_BYTES_PER_RAY
        bytes_per_launch = EXTEND_BYTES_PER_RAY * rays / max(hot_launches, 1)
    elif name == "wavefront_per_bounce_fused":
        kernel, unit, units_per_s, bytes_per_unit = SHOOT_KERNEL, "ray", rays_all / elapsed, SHOOT_BYTES_PER_RAY
        bytes_per_launch = SHOOT_BYTES_PER_RAY * rays / max(hot_launches, 1)
    else:
        slices = max(int(tot.get("pixel_slices", 1)), 1)
        kernel, unit, units_per_s = "k_megakernel", "pixel", pixels * args.steps / elapsed
        bytes_per_unit = MEGAKERNEL_BYTES_PER_PIXEL + SLICE_HANDOVER_BYTES * (slices - 1)
        bytes_per_launch = bytes_per_unit * pixels
    mean_launch_s = (hot_ms / max(hot_launches, 1)) * 1e-3
    per_launch = bytes_per_launch / mean_launch_s / 1e9 if mean_launch_s > 0 else 0.0
    achieved = units_per_s * bytes_per_unit / 1e9
    roof = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
            "formula": f"{unit}s per second of the timed region x {bytes_per_unit} B per {unit} (SURVEY.md 8d) / {HBM_PEAK_GBS:.0f} GB/s",
            "bytes_per_" + unit: bytes_per_unit, unit + "s_per_s": round(units_per_s, 1),
            "per_launch": {"achieved": round(per_launch, 3), "frac": round(per_launch / HBM_PEAK_GBS, 6), "launches": hot_launches,
                           "mean_launch_ms": round(mean_launch_s * 1e3, 5), "bytes_per_launch": round(bytes_per_launch, 1),
                           "note": "the dominant kernel alone: its algorithmic bytes per launch / its mean launch duration (hipEvents on the launch's own stream)"}}
    if name == "wavefront_per_bounce":
        # both kernels of the pipeline, each against its own algorithmic bytes; the launches of the stream lanes overlap, so the
        # per-kernel sums are device time of a lane, not wall time
        ext_s, shd_s = tot["kernel_ms"].get("wf_extend", 0.0) * 1e-3, tot["kernel_ms"].get("wf_shade", 0.0) * 1e-3
        n_pairs = max(tot["kernels"].get("wf_shade", 0) // max(args.steps, 1) * kernel_frames, 1)
        roof["pipeline"] = {
            "k_wf_extend": {"bytes_per_ray": EXTEND_BYTES_PER_RAY, "sum_launch_ms_per_frame": round(ext_s * 1e3 / kernel_frames, 3),
                            "achieved_gbs": round(EXTEND_BYTES_PER_RAY * rays / ext_s / 1e9, 3) if ext_s else None},
            "k_wf_shade": {"bytes_per_ray": SHADE_BYTES_PER_RAY, "sum_launch_ms_per_frame": round(shd_s * 1e3 / kernel_frames, 3),
                           "mean_launch_ms": round(shd_s * 1e3 / n_pairs, 5),
                           "achieved_gbs": round(SHADE_BYTES_PER_RAY * rays / shd_s / 1e9, 3) if shd_s else None,
                           "frac": round(SHADE_BYTES_PER_RAY * rays / shd_s / 1e9 / HBM_PEAK_GBS, 6) if shd_s else None},
            "stream_lanes": tot["lanes"], "launch_pairs_per_frame": n_pairs // kernel_frames,
            "kernel_durations_from": "one frame with per-launch hipEvents rendered after the timed region (events on ~3,800 launches slow the timed frames by a third)",
            "note": "the reference's shape: per sample and bounce one EXTEND + one SHADE launch, survivors compacted between bounces (wave ballot + mbcnt "
                    "prefix, one atomic per 256-thread block); interleaved sub-tiles on HIP streams overlap one lane's launch tails with the others' traversal"}
    elif name == "wavefront_per_bounce_fused":
        roof["stream_lanes"] = tot["lanes"]
        roof["schedule"] = ("one launch per sample and bounce of k_wf_finish limited to one bounce (SHOOT): intersect, shade, survivors compacted into the next queue — wave "
                            "ballot + mbcnt ranks, slots from a per-workgroup block (one device atomic per 256 rays)")
    elif name == "wavefront":
        roof["schedule"] = "one k_wf_generate + one k_wf_finish launch per frame (rt_schedule default: DESIGN.md §5)"
    if name in ("wavefront", "megakernel"):
        roof["pixel_slices"] = max(int(tot.get("pixel_slices", 1)), 1)
        roof["traversal"] = traversal_object(ATRIUM_VISITS, rays_all / elapsed, "constants of bench.py (profiles/r04_wave_stats.txt); replaced by this run's counters when the workload_sensitivity leg ran")
    roof["note"] = ("a traversal kernel's algorithmic HBM traffic is its ray-state / frame stream only (SURVEY.md §8d): the fraction is tiny by construction. Its work is "
                    "gathers from the cache-resident scene and instruction issue: see `binding` (measured VALU issue busy and vector-L1 tag look-ups) and `traffic`")
    return roof


def traversal_object(visits, rays_per_s, source):
    """SURVEY §8d's 'traversal bytes per ray' split by who serves them: node visits in the LDS-staged top of the tree against the LDS peak,
    everything else (nodes below the top, triangle records, hit shading) through the vector L1 / L2 against the L2 peak."""
    lds = visits["node_visits_in_lds_per_ray"] * NODE_BYTES
    cached = (visits["node_visits_per_ray"] - visits["node_visits_in_lds_per_ray"]) * NODE_BYTES + visits["triangle_tests_per_ray"] * TRI_BYTES + HIT_SHADING_BYTES
    return {"bytes_per_ray": round(lds + cached, 1), "from_lds_bytes_per_ray": round(lds, 1), "through_l1_l2_bytes_per_ray": round(cached, 1),
            "from_lds_gbs": round(lds * rays_per_s / 1e9, 1), "lds_peak_gbs": LDS_PEAK_GBS, "from_lds_frac": round(lds * rays_per_s / 1e9 / LDS_PEAK_GBS, 4),
            "through_l1_l2_gbs": round(cached * rays_per_s / 1e9, 1), "l2_peak_gbs": L2_PEAK_GBS, "through_l1_l2_frac": round(cached * rays_per_s / 1e9 / L2_PEAK_GBS, 4),
            "visits": visits, "visits_source": source,
            "note": "64-B nodes, 40-B triangle records, 124 B of hit shading per ray; peaks: MI355X_MICROARCH.md (LDS ~150 TB/s aggregate, L2 ~34.5 TB/s aggregate)"}


# ---- live counters ------------------------------------------------------------------------------------------------------------------
def pmc_child(args):
    """One untimed frame of the bench configuration per listed renderer: the process `rocprofv3 --pmc ... --` wraps."""
    from rtamd import abi, scenes
    from rtamd.renderer import Camera, MegakernelRenderer, Scene, WavefrontRenderer
    sd = scenes.atrium_scene(args.detail)
    scene = Scene(sd, device=0, bvh=abi.RT_BVH_SAH if args.bvh == "sah" else abi.RT_BVH_LBVH)
    cam = Camera.for_scene(sd, (args.width, args.height))
    out = {}
    for name in args.pmc_child.split(","):
        cls = MegakernelRenderer if name == "megakernel" else WavefrontRenderer
        r = cls(scene, (args.width, args.height), args.depth, args.spp)
        if name.startswith("wavefront_per_bounce"):
            r.set_schedule(finish_depth=abi.RT_SCHED_ALL_BOUNCES, fused_bounce=name.endswith("fused"))
        fr = r.render_frame(cam, want_f32=False, want_u8=False)
        out[name] = {"rays": fr.rays, "device_ms": fr.device_ms}
        r.close()
    print(json.dumps(out), flush=True)
    scene.close()
    return 0


SENSITIVITY_SCENES = ("atrium", "atrium_rotated", "atrium_tilted", "voxel_terrain")


def sensitivity_scene(name, detail):
    from rtamd import scenes
    if name == "atrium":
        return scenes.atrium_scene(detail)
    if name == "atrium_rotated":   # the bench scene's own triangles, turned 33 degrees about (0.3, 1, 0.2): nothing is axis-aligned any more
        return scenes.atrium_tilted_scene(detail, coarse=False)
    if name == "atrium_tilted":    # ... with walls, floor, slabs and roof as TWO triangles each (how a modelling tool exports them; the real Sponza's large polygons)
        return scenes.atrium_tilted_scene(detail, coarse=True)
    return scenes.voxel_scene(detail)  # the reference's second workload class (benchmark.py:18 minecraft.glb): axis-aligned unit faces, open sky


def stats_child(args):
    """One 4-spp frame per listed scene with the megakernel under RT_KERNEL_STATS=1 (the STATS instantiation's own counters go to stderr, the
    parent parses them): node visits and triangle tests per ray do not depend on the sample count."""
    from rtamd import abi
    from rtamd.renderer import Camera, MegakernelRenderer, Scene
    for name in args.stats_child.split(","):
        sd = sensitivity_scene(name, args.detail)
        scene = Scene(sd, device=0, bvh=abi.RT_BVH_SAH if args.bvh == "sah" else abi.RT_BVH_LBVH)
        r = MegakernelRenderer(scene, (args.width, args.height), args.depth, 4)
        sys.stderr.write(f"[stats child] scene {name}\n"); sys.stderr.flush()
        r.render_frame(Camera.for_scene(sd, (args.width, args.height)), want_f32=False, want_u8=False)
        r.close(); scene.close()
    return 0


def visits_per_ray(args, names):
    """{scene: {node_visits_per_ray, node_visits_in_lds_per_ray, triangle_tests_per_ray}} from the kernel's own counters, or {} when the child fails."""
    import re
    child = [sys.executable, str(REPO / "bench.py"), "--stats-child", ",".join(names), "--width", str(args.width), "--height", str(args.height),
             "--depth", str(args.depth), "--detail", str(args.detail), "--bvh", args.bvh]
    env = dict(os.environ, RT_KERNEL_STATS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RT_BENCH_FORCE_DIST"):
        env.pop(k, None)
    try:
        p = subprocess.run(child, env=env, capture_output=True, text=True, timeout=300)
    except Exception:
        return {}
    out, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.match(r"\[stats child\] scene (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"per ray: ([0-9.]+) inner \(([0-9.]+) in the top \d+ nodes\), ([0-9.]+) tri", line)
        if m and cur:
            out[cur] = {"node_visits_per_ray": float(m.group(1)), "node_visits_in_lds_per_ray": float(m.group(2)), "triangle_tests_per_ray": float(m.group(3))}
    return out


def workload_sensitivity(args, headline_value, abi, Scene, Camera, MegakernelRenderer):
    """Information only: how much of the headline is the SCENE. The generated atrium is an axis-aligned stand-in for Sponza — every large surface lies
    in a coordinate plane, the best case for axis-aligned bounding boxes quantised on a per-node grid. The same K frames with the megakernel on the
    same triangles turned out of the axes, on the two-triangle-wall variant, and on the reference's second workload class, each with the
    kernel's own per-ray counters."""
    W, H = args.width, args.height
    out = {"note": "the bench scene is the BEST case of this traversal (axis-aligned surfaces): the rotated rows are the same renderer on geometry that is not; "
                   "same resolution, spp, depth, warm-up and step count as the headline, one GPU", "renderer": "megakernel", "scenes": {}}
    visits = visits_per_ray(args, SENSITIVITY_SCENES)
    for name in SENSITIVITY_SCENES:
        entry = {}
        if name == "atrium":
            entry = {"value": headline_value, "unit": "Mrays/s", "note": "the headline configuration (megakernel row of `renderers`)"}
        else:
            sd = sensitivity_scene(name, args.detail)
            scene = Scene(sd, device=0, bvh=abi.RT_BVH_SAH if args.bvh == "sah" else abi.RT_BVH_LBVH)
            cam = Camera.for_scene(sd, (W, H))
            r = MegakernelRenderer(scene, (W, H), args.depth, args.spp)
            for _ in range(args.warmup):
                r.render_frame(cam, want_f32=False, want_u8=False)
            import torch
            torch.cuda.synchronize()
            t0, rays = time.perf_counter(), 0
            for _ in range(args.steps):
                rays += r.render_frame(cam, want_f32=False, want_u8=False).rays
            dt = time.perf_counter() - t0
            entry = {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dt / args.steps * 1e3, 3), "triangles": int(sd.n_triangles),
                     "rays_per_path": round(rays / args.steps / (W * H * args.spp), 3), "relative_to_headline_scene": round(rays / dt / 1e6 / headline_value, 3)}
            r.close(); scene.close()
        if name in visits:
            entry.update(visits[name])
        out["scenes"][name] = entry
    return out


# one rocprofv3 pass per group (counters only, no trace domains)
PMC_PASSES = (("FETCH_SIZE", "TCC_HIT_sum"), ("WRITE_SIZE", "TCC_MISS_sum"),
              ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "GRBM_GUI_ACTIVE"),
              ("SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_TRANS_F32"),
              ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_SALU"),
              ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"))
KERNELS_OF = {"megakernel": ("k_megakernel",), "wavefront": (FINISH_KERNEL,), "wavefront_per_bounce": ("k_wf_extend", "k_wf_shade"),
              "wavefront_per_bounce_fused": (SHOOT_KERNEL,)}


def live_pmc(args, renderers):
    """Counters of the renderers' kernels over ONE frame each of this very configuration, every pass in its own child process under
    rocprofv3 --pmc. Returns {kernel: {counter: sum over the kernel's launches, "launches": n}} or raises."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists():
        raise RuntimeError("rocprofv3 not found")
    child = [sys.executable, str(REPO / "bench.py"), "--pmc-child", ",".join(renderers), "--width", str(args.width), "--height", str(args.height),
             "--spp", str(args.spp), "--depth", str(args.depth), "--detail", str(args.detail), "--bvh", args.bvh]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RT_BENCH_FORCE_DIST"):
        env.pop(k, None)
    kernels = [k for r in renderers for k in KERNELS_OF[r]]
    sums = {k: {"launches": 0} for k in kernels}
    for counters in PMC_PASSES:
        with tempfile.TemporaryDirectory(prefix="rt_pmc_", dir="/tmp") as d:
            p = subprocess.run([exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=240)
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if p.returncode != 0 or not files:
                raise RuntimeError(f"rocprofv3 --pmc {' '.join(counters)} failed (rc {p.returncode}): {p.stderr[-300:]}")
            ids = {k: set() for k in kernels}
            for f in files:
                for row in csv.DictReader(open(f)):
                    for k in kernels:
                        if k in row["Kernel_Name"]:
                            sums[k][row["Counter_Name"]] = sums[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                            ids[k].add(row["Dispatch_Id"])
            for k in kernels:
                sums[k]["launches"] = max(sums[k]["launches"], len(ids[k]))
    return sums


def traffic_of(c, rays):
    n = max(int(c.get("launches", 1)), 1)
    fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0  # rocprofv3 reports both in KiB
    hit, miss = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
    return round((fetch + write) / n, 1), {
        "fetch_bytes_per_launch": round(fetch / n, 1), "write_bytes_per_launch": round(write / n, 1),
        "fetch_x2_upper_bound_per_launch": round((2 * fetch + write) / n, 1), "launches": n,
        "bytes_per_ray": round((fetch + write) / max(rays, 1), 1), "l2_hit_rate": round(hit / (hit + miss), 4) if hit and miss else None}


def binding_of(c, rays):
    """The bound that binds, measured: VALU issue busy. SQ_ACTIVE_INST_VALU counts one unit per issued VALU instruction (two for the
    8-cycle transcendentals), SQ_ACTIVE_INST_VALU2 the units that were issued as the SECOND instruction of a quad-cycle (gfx950 issues a
    full-rate instruction — fma / mul / add / mov / and / or / xor / add_u32 / lshr — of one wave beside any instruction of another
    wave); their difference is the number of quad-cycles in which the SIMD's vector ALU issued. Pinned on known instruction streams:
    profiles/r03_valu_calibration2.json (0.94-0.99 on 100 %-VALU streams of every class and mix)."""
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
    n, a, a2 = c["SQ_INSTS_VALU"], c["SQ_ACTIVE_INST_VALU"], c["SQ_ACTIVE_INST_VALU2"]
    busy = 4.0 * (a - a2) / (N_SIMDS * cycles)
    b = {"bound": "valu_issue", "frac": round(busy, 4), "measured": "4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
         "kernel_cycles_per_valu_instruction": round(cycles * N_SIMDS / n, 3), "busy_cycles_per_valu_instruction": round(4.0 * (a - a2) / n, 3),
         "dual_issued_share": round(2.0 * a2 / n, 4), "valu_instructions_per_ray": round(n / max(rays, 1), 1),
         "lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * a), 4) if c.get("SQ_THREAD_CYCLES_VALU") else None}
    typed = {t: c.get("SQ_INSTS_VALU_" + t) for t in ("FMA_F32", "MUL_F32", "ADD_F32", "INT32", "CVT", "TRANS_F32")}
    if all(v is not None for v in typed.values()):
        b["instruction_mix"] = {t.lower(): round(v / n, 4) for t, v in typed.items()}
        b["instruction_mix"]["other (cmp, cndmask, min/max, mov, logic, shifts)"] = round(1.0 - sum(typed.values()) / n, 4)
        b["salu_per_valu"] = round(c.get("SQ_INSTS_SALU", 0.0) / n, 3)
    if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        # the SECOND loaded resource (round 4): a CU's vector L1 looks up one tag per cycle; a lane's 16-byte load is one look-up
        acc = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
        b["vector_l1"] = {"tag_lookups_per_cycle_and_cu": round(acc / (N_CUS * cycles), 4), "peak_per_cycle_and_cu": 1.0, "lookups_per_ray": round(acc / max(rays, 1), 1),
                          "tag_conflict_stall_share_of_cycles": round(c.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0.0) / (N_CUS * cycles), 4),
                          "measured": "TCP_TOTAL_CACHE_ACCESSES_sum / (256 CUs x GRBM_GUI_ACTIVE / 8)"}
    b["note"] = ("frac = share of the kernel's cycles in which a SIMD's vector ALU issued an instruction (all SIMDs, whole launch, drain tail included): measured, "
                 "no instruction-cost model. The rest is cycles with no VALU issue (waits on LDS / memory / scalar code in all resident waves).")
    return b


def attach_pmc(out, args):
    """roofline.traffic (HBM-side bytes per launch of the dominant kernel) and roofline.binding (the bound that binds) of the headline
    renderer and of the per-bounce pipeline, from counters collected in this run."""
    lead = out["config"]["renderer"]
    # both one-launch renderers (they are within a few per cent of each other: which one leads differs from box to box) and the per-bounce pipelines
    want = [lead] + [n for n in ("megakernel", "wavefront", "wavefront_per_bounce", "wavefront_per_bounce_fused") if n != lead and n in out["renderers"]]
    c, source = None, None
    if not args.no_live_pmc:
        try:
            c = live_pmc(args, want)
            if args.dump_pmc:
                Path(args.dump_pmc).write_text(json.dumps(dict(c, source="bench.py --dump-pmc: rocprofv3 --pmc passes (counters only) of one frame per renderer at the bench "
                                                                            "configuration; counters summed over the kernel's launches; FETCH_SIZE / WRITE_SIZE in KiB"), indent=1))
            source = f"measured in this run: one more frame per renderer in a child process under rocprofv3 --pmc ({len(PMC_PASSES)} passes, counters only)"
        except Exception as e:  # no rocprofv3 / no counter access here: say so and use the committed profile of the same command
            source = f"live rocprofv3 --pmc unavailable ({str(e)[:160]}); "
    if c is None:
        prof = REPO / "profiles" / "r05_pmc_bench.json"
        if prof.exists():
            c = {k: v for k, v in json.loads(prof.read_text()).items() if isinstance(v, dict)}
            source = (source or "") + "committed profiles/r05_pmc_bench.json (same command: scripts/final_profile.sh)"
    for name in want:
        roof = out["renderers"][name]["roofline"]
        rays = out["renderers"][name]["rays_per_frame"]
        roof["pmc_source"] = source
        if not c:
            roof["traffic_note"] = "no counters available"
            continue
        main = c.get(roof["kernel"])
        if main and "FETCH_SIZE" in main and "WRITE_SIZE" in main:
            roof["traffic"], roof["traffic_detail"] = traffic_of(main, rays)
            roof["traffic_note"] = ("FETCH_SIZE + WRITE_SIZE of the kernel, separate --pmc passes, KiB -> bytes. MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts a wide "
                                    "coalesced 16 B/lane stream at half its bytes; scattered node / triangle gathers are of uncalibrated width, so the raw figure is "
                                    "`traffic` and the x2 bound is given beside it. Infinity-Cache hits are included (the scene is MALL-resident): this is fabric traffic.")
        if main and "SQ_ACTIVE_INST_VALU2" in main:
            roof["binding"] = binding_of(main, rays)
        if name == "wavefront_per_bounce" and c.get("k_wf_shade"):
            sh = c["k_wf_shade"]
            if "FETCH_SIZE" in sh:
                t, detail = traffic_of(sh, rays)
                roof["pipeline"]["k_wf_shade"]["traffic_per_launch"] = t
                roof["pipeline"]["k_wf_shade"]["traffic_detail"] = detail
                ms = roof["pipeline"]["k_wf_shade"].get("mean_launch_ms")
                if ms:
                    roof["pipeline"]["k_wf_shade"]["traffic_gbs"] = round(t / (ms * 1e-3) / 1e9, 1)
            if "SQ_ACTIVE_INST_VALU2" in sh:
                roof["pipeline"]["k_wf_shade"]["binding"] = binding_of(sh, rays)
    out["roofline"] = out["renderers"][lead]["roofline"]


def cpu_baseline(sd, W, H, depth, spp, renderer):
    """The CPU oracle (oracle/oracle_rt.cpp: a port of the reference's algorithm, std::thread over rows) on the
    same scene/camera/seeds at the same resolution but `spp` samples instead of 64 (rays/s does not depend on
    spp: benchmark_avg.csv:12,2,16,18 of the reference)."""
    from oracle import oracle as O
    from rtamd import abi
    osc = O.OracleScene(sd)
    cam = O.camera(W, H, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    kind = abi.RT_RENDERER_MEGAKERNEL if renderer == "megakernel" else abi.RT_RENDERER_WAVEFRONT
    cores = O.hardware_threads()
    t0 = time.perf_counter()
    _, _, rays = osc.render(cam, kind, depth, spp, use_bvh=True, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"same scene and camera, {W}x{H}, {spp} spp (of 64), depth {depth}: {rays} rays in {dt:.1f} s",
            "note": "CPU restatement of the reference (oracle/), not the reference's SYCL build"}


if __name__ == "__main__":
    sys.exit(main())
