"""BASELINE.json configs[1..4] on one GPU (4K frame of config[4]: this GPU's 1/8 tile): Mrays/s and ms/frame of both renderers."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
CASES = [("cube 512x512 16spp d10", scenes.cube_scene, {}, 512, 512, 16, 10, 1),
         ("cornell 1024x1024 64spp d10", scenes.cornell_scene, {}, 1024, 1024, 64, 10, 1),
         ("atrium 1920x1080 64spp d10", scenes.atrium_scene, {"detail": 4}, 1920, 1080, 64, 10, 1),
         ("atrium 3840x2160 256spp d16, tile 1/8", scenes.atrium_scene, {"detail": 4}, 3840, 2160, 256, 16, 8),
         ("voxel terrain 1920x1080 128spp d10", scenes.voxel_scene, {"detail": 4}, 1920, 1080, 128, 10, 1)]  # the reference's minecraft.glb row (benchmark_avg.csv:20-21)
for label, fn, kw, w, h, spp, depth, world in CASES:
    sd = fn(**kw); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (w, h))
    for cls in (WavefrontRenderer, MegakernelRenderer):
        r = cls(sc, (w, h), depth, spp)
        if world > 1: r.set_tile(0, world, 8)
        r.render_frame(cam, want_f32=False, want_u8=False)
        fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(2)), key=lambda f: f.device_ms)
        print(f"{label:40s} {cls.__name__:18s} rays {fr.rays:>12d}  {fr.device_ms:9.2f} ms  {fr.rays/fr.device_ms/1e3:8.1f} Mrays/s", flush=True)
        r.close()
    sc.close()
