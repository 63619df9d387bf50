"""How much of the extra node visits on rotated geometry is the 8-bit plane quantisation, how much the axis-aligned boxes themselves?
Host only (rt_scene_count_visits, no GPU): closest-hit walks of the PRODUCT tree with the decoded quantised child boxes, with the exact padded
bounds (what fp32 boxes would hold) and with two more bits per plane — for camera rays and two generations of diffuse-like bounce rays.
   usage: quantisation_probe.py [detail, default 4] [width, default 480]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import abi, scenes

detail = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = int(sys.argv[2]) if len(sys.argv) > 2 else 480
H = W * 9 // 16
lib = abi.load_library()


def host_scene(sd):
    c = sd.to_c(); h = C.c_void_p()
    abi.check(lib.rt_scene_create(C.byref(c), -1, abi.RT_BVH_SAH, C.byref(h)))
    return h, c


def walk(h, org, dirs, mode):
    n = org.shape[0]
    org = np.ascontiguousarray(org, np.float32); dirs = np.ascontiguousarray(dirs, np.float32)
    v, tt = C.c_uint64(0), C.c_uint64(0)
    t = np.zeros(n, np.float32); tri = np.zeros(n, np.uint32)
    abi.check(lib.rt_scene_count_visits(h, n, abi.fptr(org), abi.fptr(dirs), mode, C.byref(v), C.byref(tt), abi.fptr(t), abi.u32ptr(tri)))
    return v.value / n, tt.value / n, t, tri


def camera_rays(sd):
    cam = abi.rt_camera()
    ce = (C.c_float * 3)(*[float(v) for v in sd.camera.position]); di = (C.c_float * 3)(*[float(v) for v in sd.camera.direction])
    abi.check(lib.rt_camera_init(C.byref(cam), W, H, ce, di, float(sd.camera.focal_length)))
    p00, du, dv, c0 = (np.array(list(getattr(cam, k)), np.float32) for k in ("pixel00", "delta_u", "delta_v", "center"))
    ys, xs = np.mgrid[0:H, 0:W]
    d = p00 + xs[..., None].astype(np.float32) * du + ys[..., None].astype(np.float32) * dv - c0
    return np.broadcast_to(c0, (H * W, 3)).copy(), d.reshape(-1, 3).astype(np.float32)


def bounce(sd, org, dirs, t, tri, rng):
    """diffuse-like continuation of the rays that hit: from the hit point along (geometric normal facing the ray) + (random unit vector)"""
    hit = tri != 0xFFFFFFFF
    tw = sd.world_triangles()[tri[hit]]
    n = np.cross(tw[:, 1] - tw[:, 0], tw[:, 2] - tw[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True) + 1e-30
    d = dirs[hit]
    n = np.where((np.sum(n * d, 1) > 0)[:, None], -n, n)
    u = rng.uniform(-1, 1, size=n.shape); u /= np.linalg.norm(u, axis=1, keepdims=True)
    return (org[hit] + d * t[hit][:, None]).astype(np.float32), (n + u).astype(np.float32)


print(f"node visits / triangle tests per ray on the host walk of the product tree (detail {detail}, {W}x{H} camera rays + two bounce generations)")
print(f"{'scene':34s} {'rays':>8s} | {'8-bit planes':>16s} | {'exact boxes':>16s} | {'10-bit planes':>16s} | quantisation's share of the visits")
for name, make in (("atrium (bench scene)", lambda: scenes.atrium_scene(detail)), ("atrium_rotated (same triangles)", lambda: scenes.atrium_tilted_scene(detail, coarse=False)),
                   ("atrium_tilted (two-triangle walls)", lambda: scenes.atrium_tilted_scene(detail, coarse=True)), ("voxel terrain", lambda: scenes.voxel_scene(detail))):
    sd = make(); h, keep = host_scene(sd)
    rng = np.random.default_rng(7)
    org, dirs = camera_rays(sd)
    sets = []
    for gen in range(3):
        vq, tq, t, tri = walk(h, org, dirs, 0)
        ve, te, _, _ = walk(h, org, dirs, 1)
        v2, t2, _, _ = walk(h, org, dirs, 2)
        sets.append((org.shape[0], vq, tq, ve, te, v2, t2))
        org, dirs = bounce(sd, org, dirs, t, tri, rng)
    n = sum(s[0] for s in sets)
    # a path is one camera ray and ~7 bounce rays: weight the generations 1 : 1 : 6 (the later bounces look like the second)
    wts = np.array([1.0, 1.0, 6.0]); wts /= wts.sum()
    avg = [float(sum(w * s[k] for w, s in zip(wts, sets))) for k in range(1, 7)]
    print(f"{name:34s} {n:8d} | {avg[0]:7.2f} / {avg[1]:5.2f}  | {avg[2]:7.2f} / {avg[3]:5.2f}  | {avg[4]:7.2f} / {avg[5]:5.2f}  | {100 * (avg[0] - avg[2]) / avg[0]:5.1f} % of the node visits ({100 * (avg[0] - avg[4]) / avg[0]:.1f} % recovered by two more bits)")
    for gen, s in enumerate(sets):
        print(f"    generation {gen}: {s[0]:7d} rays  8-bit {s[1]:6.2f} / {s[2]:5.2f}   exact {s[3]:6.2f} / {s[4]:5.2f}   10-bit {s[5]:6.2f} / {s[6]:5.2f}")
    lib.rt_scene_destroy(h)
