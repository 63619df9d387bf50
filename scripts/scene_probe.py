"""Both renderers on one of the small scenes at a chosen size: SCENE=cube|cornell|triangle RES=WxH python3 scripts/scene_probe.py SPP [DEPTH]"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
sd = scenes.get_scene(os.environ.get("SCENE", "cube")); sc = Scene(sd, 0)
W, H = (int(v) for v in os.environ.get("RES", "1920x1080").split("x")); cam = Camera.for_scene(sd, (W, H))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for cls in (WavefrontRenderer, MegakernelRenderer):
    r = cls(sc, (W, H), depth, spp)
    r.render_frame(cam, want_f32=False, want_u8=False)
    fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
    print(f"{cls.__name__:20s} {fr.device_ms:8.3f} ms  {fr.rays / fr.device_ms / 1e3:8.0f} Mrays/s  {fr.rays / (W * H * spp):.2f} rays/path", flush=True)
    r.close()
