"""CPU tests of the oracle itself: it is pinned against analytic known answers (the reference has no
golden vectors: SURVEY.md §8c) and against the committed fixtures in tests/golden/."""
import math
from pathlib import Path

import numpy as np
import pytest

from rtamd import abi, scenes

GOLDEN = Path(__file__).resolve().parent / "golden"
NO_TRI = 0xFFFFFFFF


def test_xorshift_known_answers(oracle):
    # Marsaglia xorshift32 (13,17,5), src/xorshift.hpp:11-20; values computed with plain Python ints
    def ref(seed, n):
        a, out = seed, []
        for _ in range(n):
            a ^= (a << 13) & 0xFFFFFFFF
            a ^= a >> 17
            a ^= (a << 5) & 0xFFFFFFFF
            out.append(a)
        return out

    vals, st = oracle.xorshift(1, 4)
    assert ref(1, 4) == [270369, 67634689, 2647435461, 307599695]
    assert st == 307599695
    np.testing.assert_array_equal(vals, (np.array(ref(1, 4), np.uint32).astype(np.float32) * np.float32(2.0 ** -32)))
    for seed in (2463534242, 12345, 0xDEADBEEF):
        vals, st = oracle.xorshift(seed, 100)
        r = ref(seed, 100)
        assert st == r[-1]
        np.testing.assert_array_equal(vals, np.array(r, np.uint32).astype(np.float32) * np.float32(2.0 ** -32))
    # state 0 is absorbing (pixel (0,0)); the largest state converts to exactly 1.0f
    vals, st = oracle.xorshift(0, 5)
    assert st == 0 and not vals.any()
    assert np.float32(np.uint32(0xFFFFFFFF)) * np.float32(2.0 ** -32) == np.float32(1.0)


def test_half_roundtrip_matches_numpy(oracle):
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2 ** 32, 400000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x)]
    specials = np.array([0.0, -0.0, 1.0, 0.8, 65504.0, 65519.99, 65520.0, 1e-8, 2.98e-8, 2.9802322e-8, 3e-8, 6e-8, 6.1e-5,
                         6.1035156e-5, 1.0009766, 1.00048828125, 1.0014648, np.inf, -np.inf], np.float32)
    x = np.concatenate([x, specials, rng.uniform(-2, 2, 100000).astype(np.float32),
                        (rng.uniform(-1, 1, 100000) * 1e-5).astype(np.float32)])
    out, hb = oracle.half_roundtrip(x)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16)
    np.testing.assert_array_equal(hb, ref.view(np.uint16))
    np.testing.assert_array_equal(out, ref.astype(np.float32))


def test_camera_matches_float64_geometry(oracle):
    w, h = 1920, 1080
    cam = oracle.camera(w, h, (1.0, 2.0, 3.0), (0.3, -0.2, -1.0), 1.7)
    d = np.array([0.3, -0.2, -1.0]); d /= np.linalg.norm(d)
    right = np.cross(d, [0, 1, 0]); right /= np.linalg.norm(right)
    up = np.cross(right, d); up /= np.linalg.norm(up)
    aspect = w / h
    p00 = np.array([1.0, 2.0, 3.0]) - right * aspect + up + d * 1.7
    np.testing.assert_allclose(np.array(cam.pixel00), p00, rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.array(cam.delta_u), right * 2 * aspect / w, rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(cam.delta_v), -up * 2 / h, rtol=0, atol=1e-9)
    assert (cam.width, cam.height) == (w, h)


def _primary_rays(cam, w, h):
    X, Y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="xy")
    p00, du, dv, ce = (np.array(list(v), np.float32) for v in (cam.pixel00, cam.delta_u, cam.delta_v, cam.center))
    pts = (p00 + X[..., None] * du) + Y[..., None] * dv
    dirs = (pts - ce).astype(np.float32).reshape(-1, 3)
    return np.broadcast_to(ce, dirs.shape).copy(), dirs


def test_triangle_primary_hits_match_float64(oracle, scene_cache):
    sd = scene_cache("triangle")
    sc = oracle.OracleScene(sd)
    w = h = 96
    cam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    org, dirs = _primary_rays(cam, w, h)
    t, u, v, tri = sc.intersect(org, dirs, use_bvh=False)
    tw = sd.world_triangles()[0]
    v0, e1, e2 = tw[0], tw[1] - tw[0], tw[2] - tw[0]
    o, d = org.astype(np.float64), dirs.astype(np.float64)
    p = np.cross(d, e2); det = p @ e1
    tv = o - v0
    uu = np.einsum("ij,ij->i", tv, p) / det
    q = np.cross(tv, e1)
    vv = np.einsum("ij,ij->i", d, q) / det
    tt = q @ e2 / det
    inside = (uu >= 0) & (vv >= 0) & (uu + vv <= 1) & (tt > 1e-4)
    margin = np.minimum.reduce([uu, vv, 1 - uu - vv])
    clear = np.abs(margin) > 1e-5  # away from the edges both must agree exactly
    assert inside.sum() > 400
    np.testing.assert_array_equal((tri != NO_TRI)[clear], inside[clear])
    hit = (tri != NO_TRI) & inside
    np.testing.assert_allclose(t[hit], tt[hit], rtol=2e-6)
    np.testing.assert_allclose(u[hit], uu[hit], atol=2e-6)
    np.testing.assert_allclose(v[hit], vv[hit], atol=2e-6)


def test_cube_closest_t_matches_slab_test(oracle, scene_cache):
    sd = scene_cache("cube")
    sc = oracle.OracleScene(sd)
    w = h = 128
    cam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    org, dirs = _primary_rays(cam, w, h)
    t, _, _, tri = sc.intersect(org, dirs, use_bvh=False)
    c = np.array([0.05813104659318924, 0.1505535989999771, -2.920884370803833])  # node translation in cube.glb
    lo, hi = c - 1, c + 1
    o, d = org.astype(np.float64), dirs.astype(np.float64)
    with np.errstate(divide="ignore"):
        t0, t1 = (lo - o) / d, (hi - o) / d
    tn = np.minimum(t0, t1).max(1)
    tf = np.maximum(t0, t1).min(1)
    hit = (tn <= tf) & (tf > 1e-4)
    texp = np.where(tn > 1e-4, tn, tf)
    clear = np.abs(tn - tf) > 1e-4
    np.testing.assert_array_equal((tri != NO_TRI)[clear], hit[clear])
    both = hit & (tri != NO_TRI)
    assert both.sum() > 1000
    np.testing.assert_allclose(t[both], texp[both], rtol=3e-6)


def test_schlick_r0_and_dielectric_branches(oracle, scene_cache):
    sd = scene_cache("cornell")
    sc = oracle.OracleScene(sd)
    glass = [i for i, m in enumerate(sd.materials) if m.type == abi.RT_MAT_DIELECTRIC][0]
    n = np.array([[0.0, 1.0, 0.0]], np.float32)
    # normal incidence from outside: cos=1 -> reflectance = r0 = ((1-1/1.5)/(1+1/1.5))^2 = 0.04
    # seed whose first draw is > 0.04 -> refract straight through (direction unchanged)
    seed = next(s for s in ((k * 2654435761) % 2**32 for k in range(1, 100)) if oracle.xorshift(s, 1)[0][0] > 0.5)
    ok, od, oa, so = sc.scatter(glass, [[0.0, -1.0, 0.0]], n, [[0, 0]], [seed])
    assert ok[0] == 1 and np.allclose(od[0], [0, -1, 0], atol=1e-6) and np.array_equal(oa[0], [1, 1, 1])
    assert so[0] == oracle.xorshift(seed, 1)[1]  # exactly one draw consumed
    # from inside at a grazing angle: total internal reflection, NO draw consumed (short-circuit ||)
    d = np.array([[0.9, 0.43588989, 0.0]], np.float32)  # leaving through +y from inside, sin > 1/1.5
    ok, od, oa, so = sc.scatter(glass, d, n, [[0, 0]], [1234567])
    assert ok[0] == 1 and so[0] == 1234567
    np.testing.assert_allclose(od[0], [0.9, -0.43588989, 0.0], atol=1e-6)
    # reflectance probability at normal incidence is 4%: count reflections over many seeds
    seeds = np.arange(1, 20001, dtype=np.uint32) * np.uint32(2654435761)
    ok, od, _, _ = sc.scatter(glass, np.tile([[0.0, -1.0, 0.0]], (seeds.size, 1)), np.tile(n, (seeds.size, 1)),
                              np.zeros((seeds.size, 2)), seeds)
    frac = (od[:, 1] > 0).mean()
    assert abs(frac - 0.04) < 0.006


def test_material_scatter_semantics(oracle, scene_cache):
    sd = scene_cache("cornell")
    sc = oracle.OracleScene(sd)
    metal = [i for i, m in enumerate(sd.materials) if m.type == abi.RT_MAT_METALLIC][0]
    diffuse = 0
    nrm = np.array([[0.0, 1.0, 0.0]], np.float32)
    d = np.array([[0.6, -0.8, 0.0]], np.float32)
    seeds = np.arange(1, 5001, dtype=np.uint32) * np.uint32(747796405)
    N = seeds.size
    ok, od, oa, so = sc.scatter(diffuse, np.tile(d, (N, 1)), np.tile(nrm, (N, 1)), np.zeros((N, 2)), seeds)
    assert ok.all()
    # diffuse: out = n + unit vector -> |out - n| == 1, attenuation == albedo
    np.testing.assert_allclose(np.linalg.norm(od - nrm, axis=1), 1.0, atol=1e-6)
    np.testing.assert_array_equal(oa, np.tile(np.array(sd.materials[0].color, np.float32), (N, 1)))
    # three draws consumed: state advanced exactly 3 xorshift steps
    vals, st = oracle.xorshift(int(seeds[0]), 3)
    assert so[0] == st
    # metal: out = reflect + roughness * unit
    ok, od, oa, so = sc.scatter(metal, np.tile(d, (N, 1)), np.tile(nrm, (N, 1)), np.zeros((N, 2)), seeds)
    refl = np.array([0.6, 0.8, 0.0])
    np.testing.assert_allclose(np.linalg.norm(od - refl, axis=1), sd.materials[metal].roughness, atol=1e-6)
    np.testing.assert_array_equal(ok.astype(bool), od[:, 1] > 0)


def test_empty_scene_is_sky(oracle, scene_cache):
    sd = scene_cache("empty")
    sc = oracle.OracleScene(sd)
    cam = oracle.camera(32, 16, (0, 0, 0), (0, 0, -1), 1.0)
    for kind in (0, 1):
        f, b, rays = sc.render(cam, kind, 10, 3)
        assert rays == 32 * 16 * 3
        exp = np.sqrt(np.array([0.5, 0.7, 1.0], np.float32))
        # mean of 3 identical samples then sqrt; (x+x+x)/3 may differ from x by 1 ulp
        np.testing.assert_allclose(f[..., :3], np.broadcast_to(exp, f[..., :3].shape), rtol=3e-7)
        assert (f[..., 3] == 1).all() and (b[..., 3] == 255).all()
        np.testing.assert_array_equal(b[0, 0, :3], np.rint(f[0, 0, :3] * 255).astype(np.uint8))


def test_convex_diffuse_known_answer(oracle):
    """A convex diffuse object under a constant sky: every scattered ray escapes, so a hit pixel's sample
    is exactly half(albedo) * sky and a miss is sky — checks shading, fp16 storage and accumulation."""
    sb = scenes.SceneBuilder("ball")
    m = sb.add_material(scenes.Material(abi.RT_MAT_DIFFUSE, (0.8, 0.6, 0.4)))
    sb.add_instance(sb.add_mesh(*scenes.mesh_icosphere(2)), m, scenes.trs((0, 0, -3)))
    sb.sky = np.array([0.5, 0.7, 1.0], np.float32)
    sd = sb.build()
    sc = oracle.OracleScene(sd)
    w = h = 48
    cam = oracle.camera(w, h, (0, 0, 0), (0, 0, -1), 1.0)
    f, b, rays = sc.render(cam, 0, 10, 1, use_bvh=False)
    alb = np.array([0.8, 0.6, 0.4], np.float32).astype(np.float16).astype(np.float32)
    sky = np.array([0.5, 0.7, 1.0], np.float32)
    hit_val, miss_val = np.sqrt(alb * sky), np.sqrt(sky)
    px = f[..., :3].reshape(-1, 3)
    is_hit = np.all(px == hit_val, axis=1)
    is_miss = np.all(px == miss_val, axis=1)
    # the icosphere is convex up to faceting: a scattered ray can re-hit only at silhouette grazing angles
    assert (is_hit | is_miss).mean() > 0.995
    assert 0.05 < is_hit.mean() < 0.5
    assert rays == (is_miss.sum() + 2 * is_hit.sum()) + sum(
        1 for _ in range(0)) or rays >= is_miss.sum() + 2 * is_hit.sum()


@pytest.mark.parametrize("name,kw,w,h,spp", [("cube", {}, 40, 40, 3), ("cornell", {}, 40, 40, 3),
                                               ("atrium", {"detail": 1}, 48, 27, 1)])
def test_bvh_equals_brute_force(oracle, scene_cache, name, kw, w, h, spp):
    sd = scene_cache(name, **kw)
    sc = oracle.OracleScene(sd)
    cam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    f0, b0, r0 = sc.render(cam, 0, 6, spp, use_bvh=False)
    f1, b1, r1 = sc.render(cam, 0, 6, spp, use_bvh=True)
    assert r0 == r1
    np.testing.assert_array_equal(f0, f1)
    np.testing.assert_array_equal(b0, b1)


def test_megakernel_equals_wavefront_when_unclamped(oracle, scene_cache):
    """Appendix A-E3: equal seeds and no sample above 1 -> the two renderers agree exactly."""
    for name in ("triangle", "cube"):
        sd = scene_cache(name)
        sc = oracle.OracleScene(sd)
        cam = oracle.camera(56, 40, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
        for seed_mode in (abi.RT_SEED_WAVEFRONT, abi.RT_SEED_MEGAKERNEL):
            fm, bm, rm = sc.render(cam, abi.RT_RENDERER_MEGAKERNEL, 10, 5, seed_mode=seed_mode)
            fw, bw, rw = sc.render(cam, abi.RT_RENDERER_WAVEFRONT, 10, 5, seed_mode=seed_mode)
            assert rm == rw
            np.testing.assert_array_equal(fm, fw)
        # default seeds differ between the renderers (A5) -> different images
        fm, _, rm = sc.render(cam, abi.RT_RENDERER_MEGAKERNEL, 10, 5)
        fw, _, rw = sc.render(cam, abi.RT_RENDERER_WAVEFRONT, 10, 5)
        assert not np.array_equal(fm, fw)


def test_wavefront_clamps_per_sample(oracle, scene_cache):
    sd = scene_cache("cornell")  # emissive 15 -> samples above 1
    sc = oracle.OracleScene(sd)
    cam = oracle.camera(48, 48, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    fm, _, rm = sc.render(cam, 0, 10, 4, seed_mode=abi.RT_SEED_WAVEFRONT)
    fw, _, rw = sc.render(cam, 1, 10, 4, seed_mode=abi.RT_SEED_WAVEFRONT)
    assert rm == rw
    assert fw.max() <= 1.0 and fm.max() > 1.0
    assert (fw[..., :3] <= fm[..., :3] + 1e-7).all()


def test_tile_split_is_partition_invariant(oracle, scene_cache):
    from rtamd.renderer import assemble_tiles
    sd = scene_cache("cube")
    sc = oracle.OracleScene(sd)
    w, h = 40, 37  # height not a multiple of the strip
    cam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    full, _, rays = sc.render(cam, 1, 10, 2)
    for world in (2, 3):
        parts, total = [], 0
        for rank in range(world):
            f, _, r = sc.render(cam, 1, 10, 2, rank=rank, world=world, strip_rows=8)
            parts.append(f)
            total += r
        assert total == rays
        np.testing.assert_array_equal(assemble_tiles(parts, h, world, 8), full)


def test_seed_zero_pixel_is_degenerate(oracle, scene_cache):
    sd = scene_cache("cube")
    sc = oracle.OracleScene(sd)
    cam = oracle.camera(16, 16, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    f1, _, _ = sc.render(cam, 1, 10, 1)
    f8, _, _ = sc.render(cam, 1, 10, 8)
    # pixel (0,0) has seed 0: every sample is identical, so the mean equals the single sample
    np.testing.assert_allclose(f8[0, 0], f1[0, 0], rtol=3e-7)


@pytest.mark.parametrize("case", sorted(p.stem for p in GOLDEN.glob("*.npz")))
def test_oracle_reproduces_golden_fixtures(oracle, scene_cache, case):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", GOLDEN / "make_golden.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    scene, kw, w, h, spp, depth = mg.CASES[case]
    g = np.load(GOLDEN / f"{case}.npz")
    sd = scene_cache(scene, **kw)
    assert sd.n_triangles == int(g["n_triangles"])
    sc = oracle.OracleScene(sd)
    cam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    for kind, tag in ((0, "mega"), (1, "wave")):
        f, b, rays = sc.render(cam, kind, depth, spp, use_bvh=True)  # fixtures were made by brute force where small
        assert rays == int(g[f"{tag}_rays"])
        np.testing.assert_array_equal(f, g[f"{tag}_f32"])
        np.testing.assert_array_equal(b, g[f"{tag}_u8"])


def test_russian_roulette_extension_is_off_by_default_and_unbiased(oracle, scene_cache):
    """rr_start=0 is the reference's behaviour (the goldens above); turning the extension on thins the paths (far fewer
    rays) and leaves the expectation of the megakernel image unchanged. The wavefront renderer clamps every SAMPLE to
    [0,1] (src/render_wavefront.cpp:275-278), so boosted survivors clip there and its image gets slightly darker.
    (PLAN.md:23-27 lists Russian roulette as a to-do upstream.)"""
    sd = scene_cache("atrium", detail=1)
    osc = oracle.OracleScene(sd)
    w, h, spp = 48, 27, 256
    cam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    lin = lambda f: (f[..., :3].astype(np.float64) ** 2).mean()
    for kind in (0, 1):
        f0, _, r0 = osc.render(cam, kind, 10, spp)
        f00, _, r00 = osc.render(cam, kind, 10, spp, rr_start=0)
        assert r0 == r00 and np.array_equal(f0, f00)
        f1, _, r1 = osc.render(cam, kind, 10, spp, rr_start=2)
        assert r1 < 0.6 * r0
        assert not np.array_equal(f0, f1)
        if kind == 0:
            assert abs(lin(f1) - lin(f0)) < 0.02 * lin(f0)  # same mean radiance within Monte-Carlo noise
        else:
            assert 0.85 * lin(f0) < lin(f1) < 1.01 * lin(f0)
        # a start bounce at or beyond max_depth never triggers
        f2, _, r2 = osc.render(cam, kind, 10, 4, rr_start=10)
        f3, _, r3 = osc.render(cam, kind, 10, 4)
        assert r2 == r3 and np.array_equal(f2, f3)


def test_texel_division_shortcut_is_exact_for_every_byte():
    """The kernels compute texel / 255 as one Newton step on b * RN(1/255) (rt_device.h unorm8_to_float): the same float
    as the IEEE division the oracle performs, for all 256 bytes (fma emulated in float64, which holds every product
    and sum here exactly)."""
    x = np.arange(256, dtype=np.float32)
    r = np.float32(1.0) / np.float32(255.0)
    q = (x * r).astype(np.float32)
    rem = x.astype(np.float64) - q.astype(np.float64) * 255.0            # fma(-q, 255, x): exact
    assert np.array_equal(rem.astype(np.float32).astype(np.float64), rem)  # ... and representable in binary32
    got = (q.astype(np.float64) + rem * np.float64(r)).astype(np.float32)  # fma(rem, r, q)
    np.testing.assert_array_equal(got, x / np.float32(255.0))
    assert (x * r != x / np.float32(255.0)).sum() > 100  # the plain product would NOT do
