"""The voxel terrain under builder knobs (RT_SAH_TRAV_COST, RT_BVH_REINSERT, RT_BVH_COLLAPSE via the environment): frame time of both renderers and,
with RT_KERNEL_STATS=1, steps per ray.   python3 scripts/voxel_probe.py [spp]"""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sd = scenes.voxel_scene(4)
for bvh, tag in ((abi.RT_BVH_SAH, "sah"), (abi.RT_BVH_LBVH, "lbvh")):
    sc = Scene(sd, 0, bvh); info = sc.info(); cam = Camera.for_scene(sd, (1920, 1080))
    line = f"{tag} nodes {info.n_nodes} sah {info.sah_cost:.2f} maxleaf {info.max_leaf_tris}"
    for cls in (MegakernelRenderer, WavefrontRenderer):
        r = cls(sc, (1920, 1080), 10, spp)
        r.render_frame(cam, want_f32=False, want_u8=False)
        best = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
        line += f" | {cls.__name__[:4]} {best.device_ms:7.2f} ms {best.rays / best.device_ms / 1e3:8.1f} Mrays/s"
        r.close()
    print(line, flush=True)
    sc.close()
