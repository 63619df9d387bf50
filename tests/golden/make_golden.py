"""Generates tests/golden/*.npz from the CPU oracle (oracle/oracle_rt.cpp).

The reference ships no golden vectors and cannot be built or imported here, so these
fixtures pin the ORACLE's output (brute-force intersector, no BVH) at the time they were made: they
catch any later drift of the oracle or of the scene generators, and the GPU tests compare against them
too. Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from oracle import oracle as O  # noqa: E402
from rtamd import scenes  # noqa: E402

CASES = {
    # name: (scene, kwargs, W, H, spp, depth)
    "triangle_64": ("triangle", {}, 64, 64, 2, 10),
    "cube_64": ("cube", {}, 64, 64, 4, 10),
    "cornell_64": ("cornell", {}, 64, 64, 4, 10),
    "cornell_48x32_d3": ("cornell", {}, 48, 32, 3, 3),
    "atrium1_64x36": ("atrium", {"detail": 1}, 64, 36, 2, 10),
}


def main():
    out_dir = Path(__file__).resolve().parent
    for name, (scene, kw, w, h, spp, depth) in CASES.items():
        sd = scenes.get_scene(scene, **kw)
        sc = O.OracleScene(sd)
        cam = O.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
        brute = sd.n_triangles <= 200
        data = {"w": w, "h": h, "spp": spp, "depth": depth, "n_triangles": sd.n_triangles}
        for kind, tag in ((0, "mega"), (1, "wave")):
            f, b, rays = sc.render(cam, kind, depth, spp, use_bvh=not brute)
            data[f"{tag}_f32"] = f
            data[f"{tag}_u8"] = b
            data[f"{tag}_rays"] = np.uint64(rays)
        # primary-ray hit table for the centre of every pixel (no jitter): t,u,v,tri
        cx = (np.arange(w, dtype=np.float32))
        cy = (np.arange(h, dtype=np.float32))
        X, Y = np.meshgrid(cx, cy, indexing="xy")
        p00, du, dv, ce = (np.array(list(v), np.float32) for v in (cam.pixel00, cam.delta_u, cam.delta_v, cam.center))
        pts = (p00 + X[..., None] * du) + Y[..., None] * dv
        dirs = (pts - ce).astype(np.float32).reshape(-1, 3)
        orgs = np.broadcast_to(ce, dirs.shape).copy()
        t, u, v, tri = sc.intersect(orgs, dirs, use_bvh=not brute)
        data.update(hit_t=t, hit_u=u, hit_v=v, hit_tri=tri)
        np.savez_compressed(out_dir / f"{name}.npz", **data)
        print(name, "mega rays", int(data["mega_rays"]), "wave rays", int(data["wave_rays"]), "hits", int((tri != 0xFFFFFFFF).sum()))


if __name__ == "__main__":
    main()
