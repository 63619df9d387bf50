"""Upper bound on what ray re-ordering can buy: lane utilisation of the traversal kernels on PERFECTLY coherent rays (primary rays only:
max_depth 1, an 8x8 pixel tile per wave / consecutive pixels per chunk) against the full depth-10 frame. RT_KERNEL_STATS=1 must be set."""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
for depth in (1, 2, 10):
    for cls in (WavefrontRenderer, MegakernelRenderer):
        r = cls(sc, (W, H), depth, 8)
        r.render_frame(cam, want_f32=False, want_u8=False)
        print(f"--- {cls.__name__} depth {depth}", file=sys.stderr, flush=True)
        fr = r.render_frame(cam, want_f32=False, want_u8=False)
        print(f"{cls.__name__} depth {depth}: {fr.rays} rays {fr.device_ms:.2f} ms {fr.rays / fr.device_ms / 1e3:.0f} Mrays/s", file=sys.stderr, flush=True)
        r.close()
