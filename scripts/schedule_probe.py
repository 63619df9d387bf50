"""The wavefront renderer's schedules side by side on the GPU box (atrium, RES / SPP / DEPTH from the environment): device time of a frame
and Mrays/s for: one launch per frame (default), a launch per sample, per bounce EXTEND + SHADE (1 / 2 / 3 stream lanes), per bounce fused
SHOOT (1 / 2 / 3 stream lanes), and the megakernel.   usage: python3 scripts/schedule_probe.py [spp]"""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = (int(v) for v in os.environ.get("RES", "1920x1080").split("x"))
depth = int(os.environ.get("DEPTH", "10"))
tile = tuple(int(v) for v in os.environ["TILE"].split("/")) if os.environ.get("TILE") else None
sd = scenes.atrium_scene(int(os.environ.get("DETAIL", "4")))
sc = Scene(sd, 0)
cam = Camera.for_scene(sd, (W, H))
ALL = abi.RT_SCHED_ALL_BOUNCES
cases = [("megakernel", None), ("one launch per frame", {}), ("a launch per sample", dict(samples_per_launch=1))]
for lanes in (1, 2, 3):
    cases.append((f"per bounce EXTEND+SHADE, {lanes} lanes", dict(finish_depth=ALL, stream_lanes=lanes)))
for lanes in (1, 2, 3, 4):
    cases.append((f"per bounce fused SHOOT, {lanes} lanes", dict(finish_depth=ALL, fused_bounce=True, stream_lanes=lanes)))
for name, sched in cases:
    r = (MegakernelRenderer if sched is None else WavefrontRenderer)(sc, (W, H), depth, spp)
    if sched: r.set_schedule(**sched)
    if tile: r.set_tile(tile[0], tile[1], 8)
    r.render_frame(cam, want_f32=False, want_u8=False)
    best = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
    print(f"{name:40s} {best.device_ms:9.3f} ms  {best.rays / best.device_ms / 1e3:9.1f} Mrays/s  launches {best.launches}", flush=True)
    r.close()
sc.close()
