"""Both renderers on BASELINE config 5's tile as stated (4K, depth 16, 256 spp, tile 0 of 8) and on the full 4K frame (depth 10, 16 spp)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); W, H = 3840, 2160; cam = Camera.for_scene(sd, (W, H))
for cls, spp, tile in ((WavefrontRenderer, 256, True), (MegakernelRenderer, 256, True), (WavefrontRenderer, 16, False), (MegakernelRenderer, 16, False)):
    r = cls(sc, (W, H), 16 if tile else 10, spp)
    if tile: r.set_tile(0, 8, 8)
    r.render_frame(cam, want_f32=False, want_u8=False)
    fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(2)), key=lambda f: f.device_ms)
    what = "config 5 tile, 256 spp" if tile else "4K full frame, 16 spp, depth 10"
    print(f"{cls.__name__:20s} {what:34s} {fr.device_ms:8.1f} ms  {fr.rays / fr.device_ms / 1e3:.0f} Mrays/s", flush=True)
    r.close()
