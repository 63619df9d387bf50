"""The floor of the chain regime, measured: ever smaller tiles of the bench frame on one GPU — 1/8 (what each of eight GPUs gets), 1/16,
1/32, 1/64 — are ONE generation of pixel chains on ever fewer waves per SIMD (4, 2, 2 on 150 CUs, 2 on 75 CUs): the time of the small tiles is the time
of the frame's longest pixel chains with (nearly) a SIMD to themselves. With RT_KERNEL_STATS=1 the kernel reports cycles per wave step and the wave
residency (mean / max).   usage: tile_floor.py [SPP] [worlds]"""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from rtamd import scenes
from rtamd.renderer import Camera, MegakernelRenderer, Scene

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
worlds = tuple(int(v) for v in sys.argv[2].split(',')) if len(sys.argv) > 2 else (8, 16, 32, 64)
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
for world in worlds:
    for rank in (0, world // 2):
        r = MegakernelRenderer(sc, (W, H), 10, spp); r.set_tile(rank, world, 8)
        fr = r.render_frame(cam, want_f32=False, want_u8=False)
        best = min(r.render_frame(cam, want_f32=False, want_u8=False).device_ms for _ in range(3))
        pixels = r.local_rows * W
        waves = (pixels + 63) // 64
        print(f"tile {rank} of {world}: {pixels} pixels = {waves} waves ({waves / 1024:.2f} per SIMD if spread over the chip) {best:8.2f} ms  {fr.rays} rays  {fr.rays / pixels / spp:.2f} rays per path", flush=True)
        r.close()
