"""How well does ONE GPU handle 1/N of the frame (what each GPU gets in the N-GPU tile split)? ideal = 1/N of the full time.
   usage: tile_scaling.py SPP [mega|wave] [worlds, default 1,2,4,8] [frames in flight, default 1,2]
With F = 2 frames in flight (two renderers, each with its own tile buffers; rt_render_frame_begin / _end) the figure is the time per frame of a
steady stream of frames — what a render farm of N GPUs sees — not one frame's latency."""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import time

from rtamd import abi, scenes
from rtamd.renderer import Camera, MegakernelRenderer, Scene, WavefrontRenderer

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
only = sys.argv[2] if len(sys.argv) > 2 else ''
worlds = tuple(int(v) for v in sys.argv[3].split(',')) if len(sys.argv) > 3 else (1, 2, 4, 8)
flights = tuple(int(v) for v in sys.argv[4].split(',')) if len(sys.argv) > 4 else (1, 2)
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
lib = abi.load_library()
for cls in (WavefrontRenderer, MegakernelRenderer):
    if only and only.lower() not in cls.__name__.lower(): continue
    base = {}
    for world in worlds:
        for F in flights:
            rs = [cls(sc, (W, H), 10, spp) for _ in range(F)]
            for r in rs: r.set_tile(0, world, 8)
            fr = rs[0].render_frame(cam, want_f32=False, want_u8=False)
            rays, extra = fr.rays, (f" {fr.pixel_slices} slices" if cls is MegakernelRenderer else "")
            if F == 1:
                best = min(rs[0].render_frame(cam, want_f32=False, want_u8=False).device_ms for _ in range(3))
            else:  # a stream of frames, F in flight: wall time per frame over 8 frames, best of 3
                best = 1e30
                for _ in range(3):
                    n, q = 8, []
                    t0 = time.perf_counter()
                    for i in range(n):
                        if len(q) == F: q.pop(0).end_frame()
                        r = rs[i % F]
                        r.begin_frame(cam, d_f32=lib.rt_renderer_tile_f32(r.h), d_u8=lib.rt_renderer_tile_u8(r.h)); q.append(r)
                    while q: q.pop(0).end_frame()
                    best = min(best, (time.perf_counter() - t0) * 1e3 / n)
            base.setdefault(F, best)
            print(f"{cls.__name__:20s} 1/{world} of the frame, {F} in flight:{extra} {best:8.2f} ms  ideal {base[F]/world:8.2f}  efficiency {base[F]/world/best*100:5.1f}%  ({rays/best/1e3:.0f} Mrays/s)", flush=True)
            for r in rs: r.close()
