"""The per-bounce fused kernel (SHOOT: k_wf_finish limited to one bounce, one launch per sample and bounce) against the number of rays a launch holds
(the pipeline is bound by launch granularity, not by its kernels): the atrium at depth 10, 8 spp, at frame sizes from 960x540 to
7680x4320 — rays per launch, Mrays/s, and the one-launch schedule of the same renderer beside it.   usage: shoot_rate.py [spp]"""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from rtamd import abi, scenes
from rtamd.renderer import Camera, Scene, WavefrontRenderer

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sd = scenes.atrium_scene(4); sc = Scene(sd, 0)
print(f"atrium, depth 10, {spp} spp: the fused per-bounce kernel on its automatic stream lanes / on one lane, and the one-launch schedule")
print(f"{'frame':>11s} {'pixels':>10s} | {'SHOOT, lanes':>13s} {'rays/launch':>12s} {'ms':>8s} {'Mrays/s':>8s} | {'SHOOT, 1 lane':>13s} {'rays/launch':>12s} {'ms':>8s} {'Mrays/s':>8s} | {'one launch ms':>13s} {'Mrays/s':>8s}")
for W, H in ((960, 540), (1920, 1080), (2560, 1440), (3840, 2160), (5120, 2880), (7680, 4320)):
    cam = Camera.for_scene(sd, (W, H))
    row = []
    for kw in (dict(finish_depth=abi.RT_SCHED_ALL_BOUNCES, fused_bounce=True), dict(finish_depth=abi.RT_SCHED_ALL_BOUNCES, fused_bounce=True, stream_lanes=1), dict()):
        r = WavefrontRenderer(sc, (W, H), 10, spp)
        if kw: r.set_schedule(**kw)
        fr = r.render_frame(cam, want_f32=False, want_u8=False)
        best = min(r.render_frame(cam, want_f32=False, want_u8=False).device_ms for _ in range(2))
        launches = fr.kernels["wf_shoot"] or fr.kernels["wf_finish"]
        row.append((fr.stream_lanes, fr.rays / max(launches, 1), best, fr.rays / best / 1e3))
        r.close()
    (l0, rl0, ms0, v0), (l1, rl1, ms1, v1), (_, _, ms2, v2) = row
    print(f"{W:5d}x{H:<5d} {W * H:10d} | {l0:13d} {rl0:12.0f} {ms0:8.2f} {v0:8.0f} | {l1:13d} {rl1:12.0f} {ms1:8.2f} {v1:8.0f} | {ms2:13.2f} {v2:8.0f}", flush=True)
