"""The loader's image path against the REFERENCE's own decoder and resizer.

oracle/_ref/libref_stb.so is built from the reference's vendored stb_image.h / stb_image_resize2.h where they lie
under /root/reference (`make -C oracle ref`; recipe and wrapper in oracle/, binary git-ignored). It does exactly what
the reference does with an embedded glTF image: stbi_load_from_memory(..., 4) (deps/include/tiny_gltf.h:2603-2634)
and stbir_resize_uint8_srgb(..., 512, 512, 0, STBIR_RGBA) (src/image_manager.hpp:52-62). Skipped where the library
was never built."""
import ctypes as C
import io
from pathlib import Path

import numpy as np
import pytest

from rtamd import abi

REPO = Path(__file__).resolve().parent.parent
REF = REPO / "oracle" / "_ref" / "libref_stb.so"
HOST = REPO / "sycl-ray-tracer_amd" / "host" / "build" / "librt_host.so"

pytestmark = pytest.mark.skipif(not REF.exists(), reason="oracle/_ref/libref_stb.so not built (no /root/reference)")


@pytest.fixture(scope="module")
def libs(rtlib):
    ref, host = C.CDLL(str(REF)), C.CDLL(str(HOST))
    ref.ref_decode_rgba8.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.c_size_t]
    ref.ref_resize_to_layer.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    host.rth_last_error.restype = C.c_char_p
    host.rth_decode_image.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_size_t]
    host.rth_resize_to_layer.argtypes = [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
    return ref, host


def ref_decode(ref, blob: bytes):
    data = np.frombuffer(blob, np.uint8)
    w, h = C.c_int(), C.c_int()
    out = np.zeros(2048 * 2048 * 4, np.uint8)
    rc = ref.ref_decode_rgba8(abi.u8ptr(data), data.size, C.byref(w), C.byref(h), abi.u8ptr(out), out.size)
    if rc != 0:
        return None
    return out[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy()


def our_decode(host, blob: bytes):
    data = np.frombuffer(blob, np.uint8)
    w, h = C.c_uint32(), C.c_uint32()
    out = np.zeros(2048 * 2048 * 4, np.uint8)
    rc = host.rth_decode_image(abi.u8ptr(data), data.size, C.byref(w), C.byref(h), abi.u8ptr(out), out.size)
    if rc != 0:
        return None
    return out[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy()


def _picture(rng, h, w):
    """Smooth structure + texture + noise, so that DCT blocks carry low and high frequencies."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = np.stack([128 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0), 128 + 90 * np.cos((xx + yy) / 11.0),
                     (xx * 3 + yy * 5) % 256], -1)
    return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P"])
@pytest.mark.parametrize("size", [(67, 91), (1, 1), (64, 64)])
def test_png_decoder_equals_the_references(libs, mode, size):
    from PIL import Image
    ref, host = libs
    rng = np.random.default_rng(hash((mode, size)) & 0xFFFF)
    img = rng.integers(0, 256, size + (4,), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img, "RGBA").convert(mode).save(buf, format="PNG")
    a, r = our_decode(host, buf.getvalue()), ref_decode(ref, buf.getvalue())
    assert a is not None and r is not None
    np.testing.assert_array_equal(a, r)


@pytest.mark.parametrize("name,shape", [("same", (512, 512)), ("half", (1024, 1024)), ("wide", (512, 1024)), ("odd", (200, 300)),
                                        ("up", (100, 64)), ("tall", (1000, 700)), ("tiny", (5, 3)), ("one", (1, 1)), ("near", (511, 513))])
@pytest.mark.parametrize("alpha", [False, True])
def test_texture_baker_follows_the_references_resize(libs, name, shape, alpha):
    """512x512 layers as the reference bakes them: identical for 512x512 sources; elsewhere at most 1 LSB away in under
    1.5 % of the values (stb's table-driven float->sRGB8 step is not restated; everything else is)."""
    ref, host = libs
    rng = np.random.default_rng(len(name) * 7 + alpha)
    img = np.concatenate([_picture(rng, *shape), np.full(shape + (1,), 255, np.uint8)], -1)
    if name in ("half", "odd"):
        img[..., :3] = rng.integers(0, 256, shape + (3,), dtype=np.uint8)  # worst case: white noise
    if alpha:
        img[..., 3] = rng.integers(0, 256, shape, dtype=np.uint8)
        img[: max(1, shape[0] // 4), :, 3] = 0  # a fully transparent band
    img = np.ascontiguousarray(img)
    a, r = np.zeros((512, 512, 4), np.uint8), np.zeros((512, 512, 4), np.uint8)
    host.rth_resize_to_layer(abi.u8ptr(img), shape[1], shape[0], abi.u8ptr(a))
    assert ref.ref_resize_to_layer(abi.u8ptr(img), shape[1], shape[0], abi.u8ptr(r)) == 0
    d = np.abs(a.astype(np.int32) - r.astype(np.int32))
    if shape == (512, 512):
        assert d.max() == 0
    else:
        assert d.max() <= 1 and (d > 0).mean() < 0.015


JPEG_CASES = [  # (name, size, PIL save options)
    ("444", (64, 64), dict(quality=90, subsampling=0)),
    ("422", (77, 53), dict(quality=85, subsampling=1)),
    ("420", (131, 97), dict(quality=75, subsampling=2)),
    ("420_q30", (200, 120), dict(quality=30, subsampling=2)),
    ("420_q100", (48, 80), dict(quality=100, subsampling=2)),
    ("tiny", (1, 1), dict(quality=90)),
    ("thin", (3, 40), dict(quality=90, subsampling=2)),
    ("one_col_chroma", (17, 2), dict(quality=90, subsampling=2)),
    ("optimised", (120, 90), dict(quality=80, optimize=True)),
    ("restart_rows", (150, 150), dict(quality=80, subsampling=2, restart_marker_rows=1)),
    ("restart_blocks", (150, 150), dict(quality=80, subsampling=0, restart_marker_blocks=3)),
    ("progressive_420", (160, 120), dict(quality=80, subsampling=2, progressive=True)),
    ("progressive_444", (95, 61), dict(quality=92, subsampling=0, progressive=True)),
    ("progressive_restart", (130, 130), dict(quality=70, subsampling=1, progressive=True, restart_marker_rows=2)),
    ("big", (512, 512), dict(quality=88, subsampling=2)),
]


@pytest.mark.parametrize("name,size,opts", JPEG_CASES, ids=[c[0] for c in JPEG_CASES])
@pytest.mark.parametrize("grey", [False, True])
def test_jpeg_decoder_equals_the_references(libs, name, size, opts, grey):
    """Baseline, optimised-Huffman, restart-interval and progressive JPEGs, colour (4:4:4 / 4:2:2 / 4:2:0) and greyscale:
    every texel equals what the reference's stb_image produces (same IDCT, chroma upsampling and YCbCr arithmetic)."""
    from PIL import Image
    ref, host = libs
    rng = np.random.default_rng(len(name) + 100 * grey)
    h, w = size[1], size[0]
    img = Image.fromarray(_picture(rng, h, w), "RGB")
    if grey:
        img = img.convert("L")
        opts = {k: v for k, v in opts.items() if k != "subsampling"}
    buf = io.BytesIO()
    img.save(buf, format="JPEG", **opts)
    a, r = our_decode(host, buf.getvalue()), ref_decode(ref, buf.getvalue())
    assert r is not None
    assert a is not None, host.rth_last_error()
    assert a.shape == (h, w, 4)
    np.testing.assert_array_equal(a, r)
    # and both are the picture PIL decodes, up to the usual decoder-to-decoder differences
    pil = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")).astype(np.int32)
    assert np.abs(a[..., :3].astype(np.int32) - pil).mean() < (1.5 if min(w, h) >= 16 else 8.0)


def test_undecodable_images_are_errors_in_both(libs):
    ref, host = libs
    for blob in (b"", b"\xff\xd8\xff", b"\xff\xd8\xff\xe0\x00\x10JFIF\x00" + b"\x00" * 64, b"GIF89a" + b"\x00" * 32, b"\x89PNG\r\n\x1a\n" + b"\x00" * 16):
        assert our_decode(host, blob) is None


def _adam7_png(img: np.ndarray, depth16: bool = False) -> bytes:
    """An interlaced (Adam7) PNG of an (h, w, c) uint8 image, c in 1 (grey) / 3 (RGB) / 4 (RGBA), written by hand (PIL writes none): seven
    reduced images, scanlines with filter types 0-4 in turn (each pass starts over with an all-zero previous line)."""
    import struct
    import zlib
    h, w, c = img.shape
    bpp = c * (2 if depth16 else 1)
    raw = bytearray()
    for k, (x0, y0, dx, dy) in enumerate(((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))):
        sub = img[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        if depth16:
            sub = np.stack([sub, (sub.astype(np.int32) * 7 % 256).astype(np.uint8)], -1).reshape(sub.shape[0], sub.shape[1], -1)  # high byte = the picture
        prev = np.zeros(sub.shape[1] * bpp, np.int32)
        for j, row in enumerate(sub.reshape(sub.shape[0], -1).astype(np.int32)):
            f = (j + k) % 5
            a = np.concatenate([np.zeros(bpp, np.int32), row[:-bpp]])
            cc = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
            if f == 0: pred = 0
            elif f == 1: pred = a
            elif f == 2: pred = prev
            elif f == 3: pred = (a + prev) // 2
            else:
                pa, pb, pc = np.abs(prev - cc), np.abs(a - cc), np.abs(a + prev - 2 * cc)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, cc))
            raw.append(f)
            raw += ((row - pred) % 256).astype(np.uint8).tobytes()
            prev = row
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    ihdr = struct.pack(">IIBBBBB", w, h, 16 if depth16 else 8, {1: 0, 3: 2, 4: 6}[c], 0, 0, 1)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b"")


@pytest.mark.parametrize("w,h,c,depth16", [(1, 1, 3, False), (2, 3, 4, False), (5, 7, 1, False), (8, 8, 3, False), (9, 9, 4, False), (33, 17, 3, False),
                                           (64, 40, 4, False), (13, 21, 3, True), (30, 5, 4, True), (3, 1, 1, False), (1, 6, 3, False)])
def test_interlaced_png_decodes_like_the_references(libs, w, h, c, depth16):
    """Adam7-interlaced PNGs (the reference's stb_image reads them, deps/include/stb_image.h:4860-4866; rounds 1-4 refused them):
    sizes with empty passes, every filter type, grey / RGB / RGBA, 8 and 16 bit — texel for texel what the reference's decoder returns."""
    ref, host = libs
    rng = np.random.default_rng(w * 100 + h + c)
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    blob = _adam7_png(img, depth16)
    a, r = our_decode(host, blob), ref_decode(ref, blob)
    assert r is not None and r.shape == (h, w, 4)
    assert a is not None, host.rth_last_error()
    np.testing.assert_array_equal(a, r)
    want = np.concatenate([np.repeat(img, 3, -1) if c == 1 else img[..., :3], img[..., 3:4] if c == 4 else np.full((h, w, 1), 255, np.uint8)], -1)
    np.testing.assert_array_equal(a, want)


@pytest.mark.parametrize("transform", [2, 0, 1])
@pytest.mark.parametrize("opts", [dict(quality=90), dict(quality=75, progressive=True), dict(quality=85, optimize=True)])
def test_four_component_jpeg_decodes_like_the_references(libs, transform, opts):
    """Adobe four-component JPEGs (stb_image reads them; rounds 1-4 refused them): plain CMYK as libjpeg writes a CMYK picture
    (APP14 transform 0), the same data declared YCCK (transform 2) and with an unknown transform (the fourth channel ignored) — every texel
    what the reference's stb_image returns."""
    from PIL import Image
    ref, host = libs
    rng = np.random.default_rng(transform)
    img = Image.fromarray(_picture(rng, 40, 56), "RGB").convert("CMYK")
    buf = io.BytesIO()
    img.save(buf, format="JPEG", **opts)
    blob = bytearray(buf.getvalue())
    at = blob.find(b"Adobe")
    assert at > 0 and blob[at + 11] == 0  # APP14: "Adobe", version (2), flags0 (2), flags1 (2), transform
    blob[at + 11] = transform
    a, r = our_decode(host, bytes(blob)), ref_decode(ref, bytes(blob))
    assert r is not None and r.shape == (40, 56, 4)
    assert a is not None, host.rth_last_error()
    np.testing.assert_array_equal(a, r)


def test_glb_with_jpeg_textures_bakes_the_references_layers(libs, scene_cache, tmp_path):
    """End to end through the C++ loader: a GLB whose textures are 640x640 JPEGs -> decoded and baked to 512x512 layers;
    each layer is what the reference's stbi_load_from_memory + stbir_resize_uint8_srgb make of the same bytes."""
    import json
    import struct
    from rtamd.glb_export import export_glb
    from test_host_cpp import CppScene
    ref, host = libs
    host.rth_load_glb.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    host.rth_scene_free.argtypes = [C.c_void_p]
    host.rth_scene_desc.argtypes = [C.c_void_p, C.POINTER(abi.rt_scene_desc)]
    host.rth_scene_camera.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    sd = scene_cache("atrium", detail=1)
    glb = tmp_path / "jpeg.glb"
    export_glb(sd, glb, jpeg_quality=85, texture_size=640)
    cs = CppScene(host, glb)
    n = cs.c.n_layers
    assert n == sd.textures.shape[0] and n >= 8
    layers = np.ctypeslib.as_array(cs.c.textures, (n, 512, 512, 4))
    raw = glb.read_bytes()
    jlen = struct.unpack_from("<I", raw, 12)[0]
    js = json.loads(raw[20:20 + jlen])
    bin0 = 20 + jlen + 8
    for i, im in enumerate(js["images"]):
        assert im["mimeType"] == "image/jpeg"
        v = js["bufferViews"][im["bufferView"]]
        blob = raw[bin0 + v["byteOffset"]: bin0 + v["byteOffset"] + v["byteLength"]]
        dec = ref_decode(ref, blob)
        assert dec.shape == (640, 640, 4)
        want = np.zeros((512, 512, 4), np.uint8)
        assert ref.ref_resize_to_layer(abi.u8ptr(np.ascontiguousarray(dec)), 640, 640, abi.u8ptr(want)) == 0
        d = np.abs(layers[i].astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.015


def test_png_written_by_the_host_adapter_reads_back_in_the_references_decoder(libs, tmp_path):
    """out.png (src/util.hpp:27 writes it with stbi_write_png): the file this build writes decodes, in the reference's
    own stb_image, to exactly the RGBA8 frame that was handed to the writer."""
    ref, host = libs
    host.rth_write_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
    rng = np.random.default_rng(9)
    for h, w in ((1, 1), (37, 53), (270, 480)):
        img = np.ascontiguousarray(rng.integers(0, 256, (h, w, 4), dtype=np.uint8))
        img[..., 3] = 255
        path = tmp_path / f"o_{w}x{h}.png"
        assert host.rth_write_png(str(path).encode(), w, h, abi.u8ptr(img)) == 0
        np.testing.assert_array_equal(ref_decode(ref, path.read_bytes()), img)
