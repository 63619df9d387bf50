"""Frame rate with the image handed back in HOST buffers (rt_render_frame: device render + D2H copies) next to the
device-resident rate bench.py reports (rt_render_frame_device)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
for cls in (WavefrontRenderer, MegakernelRenderer):
    r = cls(sc, (W, H), 10, spp)
    r.render_frame(cam)
    for label, kw in (("device-resident", dict(want_f32=False, want_u8=False)), ("host u8 (CLI)", dict(want_f32=False, want_u8=True)),
                      ("host f32+u8", dict(want_f32=True, want_u8=True))):
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); fr = r.render_frame(cam, **kw); best = min(best, time.perf_counter() - t)
        print(f"{cls.__name__:20s} {label:16s} {best*1e3:8.2f} ms/frame  {fr.rays/best/1e6:8.1f} Mrays/s", flush=True)
    r.close()
