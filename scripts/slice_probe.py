"""Pixel slices (rt_schedule::pixel_slices) on 1/N tiles of the bench frame: ms, Mrays/s, the slice count that ran and a digest of the fp32 tile
(every setting must render the same bits).   usage: slice_probe.py [worlds, e.g. 1,2,4,8] [settings, e.g. 0,-1,2,3,4,8] [spp] [mega|wave]"""
import hashlib
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from rtamd import scenes
from rtamd.renderer import Camera, MegakernelRenderer, Scene, WavefrontRenderer

worlds = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
settings = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,-1").split(",")]
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cls = WavefrontRenderer if len(sys.argv) > 4 and sys.argv[4].startswith("wave") else MegakernelRenderer
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
for world in worlds:
    for n in settings:
        r = cls(sc, (W, H), 10, spp); r.set_schedule(pixel_slices=n); r.set_tile(0, world, 8)
        fr = r.render_frame(cam, want_f32=True, want_u8=False)
        digest = hashlib.sha1(fr.rgba_f32.tobytes()).hexdigest()[:12]
        best = min(r.render_frame(cam, want_f32=False, want_u8=False).device_ms for _ in range(3))
        print(f"1/{world} pixel_slices {n:2d} -> {fr.pixel_slices} slices {best:8.2f} ms  {fr.rays / best / 1e3:7.0f} Mrays/s  rays {fr.rays}  frame {digest}", flush=True)
        r.close()
