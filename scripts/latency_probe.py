"""Per-ray latency of ONE pixel chain at different chip loads: render 1/world of the frame (megakernel, no refill possible
when pixels < lanes) and divide the frame time by the longest chain (spp * depth rays)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer
spp = 16
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
for world in (8, 16, 32, 64, 128):
    r = MegakernelRenderer(sc, (W, H), 10, spp); r.set_tile(0, world, 8)
    r.render_frame(cam, want_f32=False, want_u8=False)
    best = min(r.render_frame(cam, want_f32=False, want_u8=False).device_ms for _ in range(3))
    npx = r.local_rows * W
    print(f"1/{world}: {npx} px = {npx//64} waves ({npx/64/1024:.2f} per SIMD): {best:7.2f} ms -> {best*1e3/(spp*10):6.1f} us per ray of the longest chain", flush=True)
    r.close()
