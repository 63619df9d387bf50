"""Fast A/B timing on the GPU box: both renderers, atrium 1080p, few spp. Prints Mrays/s per renderer."""
import sys, time, os
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
import numpy as np
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = sys.argv[2] if len(sys.argv) > 2 else "both"
detail = int(os.environ.get("DETAIL", "4"))
W, H = (int(v) for v in os.environ.get("RES", "1920x1080").split("x"))
sd = scenes.atrium_scene(detail)
for bvh, bname in ((abi.RT_BVH_SAH, "sah"),) + (((abi.RT_BVH_LBVH, "lbvh"),) if os.environ.get("LBVH") else ()):
    sc = Scene(sd, 0, bvh)
    cam = Camera.for_scene(sd, (W, H))
    for cls in (WavefrontRenderer, MegakernelRenderer):
        if which != "both" and which not in cls.__name__.lower(): continue
        r = cls(sc, (W, H), int(os.environ.get("DEPTH", "10")), spp)
        r.render_frame(cam, want_f32=False, want_u8=False)
        best = None
        for _ in range(3):
            fr = r.render_frame(cam, want_f32=False, want_u8=False)
            if best is None or fr.device_ms < best.device_ms: best = fr
        print(f"{cls.__name__:20s} {bname} spp={spp} rays={best.rays} {best.device_ms:9.3f} ms  {best.rays/best.device_ms/1e3:9.1f} Mrays/s", flush=True)
        r.close()
    sc.close()
