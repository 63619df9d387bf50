"""How well does ONE GPU handle 1/N of the frame (what each GPU gets in the N-GPU tile split)? ideal = 1/N of the full time."""
import sys, os
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
only = sys.argv[2] if len(sys.argv) > 2 else ''
for cls in (WavefrontRenderer, MegakernelRenderer):
    if only and only.lower() not in cls.__name__.lower(): continue
    base = None
    for world in (tuple(int(v) for v in sys.argv[3].split(',')) if len(sys.argv) > 3 else (1, 2, 4, 8)):
        r = cls(sc, (W, H), 10, spp); r.set_tile(0, world, 8)
        r.render_frame(cam, want_f32=False, want_u8=False)
        best = min(r.render_frame(cam, want_f32=False, want_u8=False).device_ms for _ in range(3))
        rays = r.render_frame(cam, want_f32=False, want_u8=False).rays
        if base is None: base = best
        print(f"{cls.__name__:20s} 1/{world} of the frame: {best:8.2f} ms  ideal {base/world:8.2f}  efficiency {base/world/best*100:5.1f}%  ({rays/best/1e3:.0f} Mrays/s)", flush=True)
        r.close()
