"""GPU parity of the kernel experiments that are compile-time variants of the traversal: the variant library is
built beside the product library (`make -C sycl-ray-tracer_amd/csrc experiments`, part of __graft_entry__.build()) and loaded in a
child process through RT_MI355X_LIB; its frames are compared BIT FOR BIT with the CPU oracle here. The product never loads it."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from rtamd import abi

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
BUILD = REPO / "sycl-ray-tracer_amd" / "csrc" / "build"
# variant -> (library, environment of the child process)
#   postpone: RT_POSTPONE=1 traversal (speculative inner steps for lanes at a leaf)
#   ts      : k_megakernel_ts, seven traversal waves and one shading wave per workgroup exchanging pixel slots through LDS queues
#             (rt_kernel_ts.h; compiled into this library only, selected with RT_MEGA_TS=1): a pixel's rays are traced and shaded one
#             after the other by DIFFERENT lanes, so frame, unorm8 image and ray count must still equal the oracle's MegakernelRenderer
#   ordered : RT_ORDERED=1, children visited in stored (builder-sorted) order by the ray's direction sign instead of sorted by entry distance
VARIANTS = {"postpone": (BUILD / "librt_exp_postpone.so", {}), "ts": (BUILD / "librt_exp_ts.so", {"RT_MEGA_TS": "1"}),
            "ordered": (BUILD / "librt_exp_ordered.so", {})}
# beyond the two base frames: tiny frames (fewer pixels than one workgroup has slots), frames that are not a multiple of the 8x8 tiles,
# an odd tile of a split, Russian roulette, max_depth 1 — megakernel only
EXTRA = [("cube", {}, (1, 1), 5, 2, None, 0), ("cube", {}, (37, 21), 1, 3, None, 0), ("cornell", {}, (333, 187), 6, 2, (1, 3), 0),
         ("cornell", {}, (96, 64), 8, 4, None, 2)]

CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
out = {}
for name, kw, (w, h), depth, spp in (("cornell", {}, (96, 64), 8, 4), ("atrium", {"detail": 1}, (192, 108), 10, 3)):
    sd = scenes.get_scene(name, **kw)
    sc = Scene(sd, 0)
    cam = Camera.for_scene(sd, (w, h))
    for cls in (MegakernelRenderer, WavefrontRenderer):
        r = cls(sc, (w, h), depth, spp)
        fr = r.render_frame(cam)
        out[f"{name}_{cls.__name__}_f32"] = fr.rgba_f32
        out[f"{name}_{cls.__name__}_u8"] = fr.rgba_u8
        out[f"{name}_{cls.__name__}_rays"] = np.array([fr.rays], dtype=np.uint64)
        r.close()
    # closest hits of a ray batch: the traversal alone
    rng = np.random.default_rng(7)
    lo, hi = sd.world_triangles().reshape(-1, 3).min(0), sd.world_triangles().reshape(-1, 3).max(0)
    org = (lo + (hi - lo) * rng.random((4096, 3))).astype(np.float32)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    t, u, v, tri = sc.intersect(org, d)
    out[f"{name}_hits_t"], out[f"{name}_hits_tri"], out[f"{name}_org"], out[f"{name}_dir"] = t, tri, org, d
    sc.close()
import json
for i, (name, kw, (w, h), depth, spp, tile, rr) in enumerate(json.loads(sys.argv[2])):
    sd = scenes.get_scene(name, **kw)
    sc = Scene(sd, 0)
    r = MegakernelRenderer(sc, (w, h), depth, spp)
    if tile: r.set_tile(tile[0], tile[1], 8)
    if rr: r.set_russian_roulette(rr)
    for rep in range(2):  # the same renderer twice
        fr = r.render_frame(Camera.for_scene(sd, (w, h)))
        out[f"extra{i}_{rep}_f32"], out[f"extra{i}_{rep}_u8"], out[f"extra{i}_{rep}_rays"] = fr.rgba_f32, fr.rgba_u8, np.array([fr.rays], dtype=np.uint64)
    r.close(); sc.close()
np.savez(sys.argv[1], **out)
'''


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_traversal_variant_equals_the_oracle(rtlib, oracle, scene_cache, tmp_path, variant):
    import json
    lib, env = VARIANTS[variant]
    if not lib.exists():
        pytest.skip(f"{lib.name} not built (make -C sycl-ray-tracer_amd/csrc experiments)")
    npz = tmp_path / "frames.npz"
    p = subprocess.run([sys.executable, "-c", CHILD, str(npz), json.dumps(EXTRA)], cwd=REPO, env=dict(os.environ, RT_MI355X_LIB=str(lib), **env),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    got = np.load(npz)
    for name, kw, (w, h), depth, spp in (("cornell", {}, (96, 64), 8, 4), ("atrium", {"detail": 1}, (192, 108), 10, 3)):
        sd = scene_cache(name, **kw)
        osc = oracle.OracleScene(sd)
        ocam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
        for cls, kind in (("MegakernelRenderer", abi.RT_RENDERER_MEGAKERNEL), ("WavefrontRenderer", abi.RT_RENDERER_WAVEFRONT)):
            f, b, rays = osc.render(ocam, kind, depth, spp, use_bvh=(name == "atrium"))
            assert int(got[f"{name}_{cls}_rays"][0]) == rays, f"{variant} {name} {cls} ray count"
            np.testing.assert_array_equal(got[f"{name}_{cls}_f32"], f, err_msg=f"{variant} {name} {cls} fp32 frame")
            np.testing.assert_array_equal(got[f"{name}_{cls}_u8"], b, err_msg=f"{variant} {name} {cls} unorm8 image")
        t, u, v, tri = osc.intersect(got[f"{name}_org"], got[f"{name}_dir"], use_bvh=False)  # brute force on the CPU
        np.testing.assert_array_equal(got[f"{name}_hits_tri"], tri, err_msg=f"{variant} {name} closest triangles")
        np.testing.assert_array_equal(got[f"{name}_hits_t"], t, err_msg=f"{variant} {name} hit distances")
    for i, (name, kw, (w, h), depth, spp, tile, rr) in enumerate(EXTRA):
        sd = scene_cache(name, **kw)
        ocam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
        tk = dict(rank=tile[0], world=tile[1], strip_rows=8) if tile else {}
        f, b, rays = oracle.OracleScene(sd).render(ocam, abi.RT_RENDERER_MEGAKERNEL, depth, spp, rr_start=rr, **tk)
        for rep in range(2):
            assert int(got[f"extra{i}_{rep}_rays"][0]) == rays, f"{variant} extra case {i} rep {rep}: ray count"
            np.testing.assert_array_equal(got[f"extra{i}_{rep}_f32"], f, err_msg=f"{variant} extra case {i} rep {rep}")
            np.testing.assert_array_equal(got[f"extra{i}_{rep}_u8"], b)
