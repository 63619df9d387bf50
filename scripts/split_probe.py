"""Pre-splitting of large triangles in the SAH builder (RT_BVH_SPLIT_BUDGET = 0 / 0.3): build time, tree cost, frame time and — with RT_KERNEL_STATS=1 —
node visits and triangle tests per ray, on the tilted atrium with two-triangle walls (where it applies) and on the bench's atrium (where it must not).
   python3 scripts/split_probe.py [spp]"""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, sd in (("tilted atrium, two-triangle walls", scenes.atrium_tilted_scene(4)), ("tilted atrium, tessellated walls", scenes.atrium_tilted_scene(4, coarse=False)),
                 ("bench atrium", scenes.atrium_scene(4))):
    for budget in ("0", "0.3"):
        os.environ["RT_BVH_SPLIT_BUDGET"] = budget
        t0 = time.perf_counter()
        sc = Scene(sd, 0, abi.RT_BVH_SAH)
        dt = time.perf_counter() - t0
        info = sc.info()
        cam = Camera.for_scene(sd, (1920, 1080))
        line = f"{name:34s} budget {budget:4s} build {dt * 1e3:6.0f} ms  records {info.n_leaf_records:7d} (+{info.n_leaf_records - info.n_triangles}, {info.n_split_triangles} split)  sah {info.sah_cost:6.2f}"
        for cls in (MegakernelRenderer, WavefrontRenderer):
            r = cls(sc, (1920, 1080), 10, spp)
            r.render_frame(cam, want_f32=False, want_u8=False)
            sys.stderr.flush()
            best = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
            line += f" | {cls.__name__[:4]} {best.device_ms:7.2f} ms {best.rays / best.device_ms / 1e3:7.1f} Mrays/s"
            r.close()
        print(line, flush=True)
        sc.close()
