"""Cornell box 1024x1024, 64 spp, depth 10: both renderers (RT_KERNEL_STATS=1 prints their wave statistics)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
sd = scenes.cornell_scene(); sc = Scene(sd, 0); W = H = 1024; cam = Camera.for_scene(sd, (W, H))
for cls in (WavefrontRenderer, MegakernelRenderer):
    r = cls(sc, (W, H), 10, int(sys.argv[1]) if len(sys.argv) > 1 else 64)
    r.render_frame(cam, want_f32=False, want_u8=False)
    fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(2)), key=lambda f: f.device_ms)
    print(f"{cls.__name__:20s} {fr.device_ms:8.2f} ms  {fr.rays / fr.device_ms / 1e3:.0f} Mrays/s", flush=True)
    r.close()
