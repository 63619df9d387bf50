"""Developer tool (uses the oracle as the checker, hence under tests/): prints where GPU and oracle differ."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
import numpy as np
from oracle import oracle as O
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
sd = scenes.get_scene("cornell")
gs, osc = Scene(sd, 0), O.OracleScene(sd)
w = h = 64
cam = Camera.for_scene(sd, (w, h)); ocam = O.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
np.set_printoptions(precision=9, floatmode="unique")
for spp in (1, 2, 3, 4):
    r = MegakernelRenderer(gs, (w, h), 10, spp)
    fr = r.render_frame(cam); r.close()
    f, b, rays = osc.render(ocam, 0, 10, spp, use_bvh=False)
    bad = np.argwhere((fr.rgba_f32 != f).any(-1))
    print("spp", spp, "rays", fr.rays, rays, "bad", len(bad))
    for (y, x) in bad[:6]:
        print("  px", x, y, "gpu", fr.rgba_f32[y, x, :3], (fr.rgba_f32[y, x, :3].astype(np.float64)**2*spp), "cpu", f[y, x, :3], (f[y, x, :3].astype(np.float64)**2*spp))
