import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer
sd = scenes.atrium_tilted_scene(4)
for budget, floor in (("0.3", "0.015625"), ("1.0", "0.015625"), ("1.0", "0.0078"), ("2.0", "0.0039"), ("0.3", "0.0078")):
    os.environ["RT_BVH_SPLIT_BUDGET"], os.environ["RT_BVH_SPLIT_FLOOR"] = budget, floor
    sc = Scene(sd, 0, abi.RT_BVH_SAH); info = sc.info()
    r = MegakernelRenderer(sc, (1920, 1080), 10, 16); cam = Camera.for_scene(sd, (1920, 1080))
    r.render_frame(cam, want_f32=False, want_u8=False)
    best = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
    print(f"budget {budget} floor {floor}: records +{info.n_leaf_records - info.n_triangles} ({info.n_split_triangles} split) sah {info.sah_cost:.2f}  {best.device_ms:.2f} ms {best.rays / best.device_ms / 1e3:.0f} Mrays/s", flush=True)
    r.close(); sc.close()
