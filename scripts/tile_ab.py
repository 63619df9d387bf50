"""A/B of run-time knobs on 1/N tiles of the bench frame (atrium 1080p, 64 spp, depth 10): ms per tile and a digest of the tile's fp32
pixels (every variant must render the same bits).
   usage: tile_ab.py <renderer: mega|wave> <worlds, e.g. 8,4,1> <label:ENV=V,ENV2=V ...>..."""
import hashlib
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from rtamd import scenes
from rtamd.renderer import Camera, MegakernelRenderer, Scene, WavefrontRenderer

which, worlds = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
variants = []
for a in sys.argv[3:]:
    label, _, envs = a.partition(":")
    variants.append((label, dict(e.split("=") for e in envs.split(",") if e)))
if any(env for _, env in variants):  # knobs exist in the developer build only (csrc/rt_knobs.h)
    os.environ.setdefault("RT_MI355X_LIB", str(REPO / "sycl-ray-tracer_amd" / "csrc" / "build" / "librt_mi355x_dev.so"))
spp = int(os.environ.get("SPP", "64"))
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
cls = MegakernelRenderer if which.startswith("mega") else WavefrontRenderer
for world in worlds:
    for label, env in variants:
        for k, v in env.items(): os.environ[k] = v
        r = cls(sc, (W, H), 10, spp); r.set_tile(0, world, 8)
        for k in env: del os.environ[k]
        fr = r.render_frame(cam, want_f32=True, want_u8=False)
        digest = hashlib.sha1(fr.rgba_f32.tobytes()).hexdigest()[:12]
        best = min(r.render_frame(cam, want_f32=False, want_u8=False).device_ms for _ in range(3))
        print(f"1/{world} {label:24s} {best:8.2f} ms  {fr.rays / best / 1e3:7.0f} Mrays/s  rays {fr.rays}  frame {digest}", flush=True)
        r.close()
