"""Slice geometries of the megakernel on 1/N tiles of the bench frame, through a developer build (RT_DEVELOPER_KNOBS: RT_MEGA_SLICE_BOUNDS):
   usage: RT_MI355X_LIB=<librt_var_dev.so> slice_sweep.py <worlds, e.g. 1,2> <bounds ; bounds ; ...>   e.g. "48,60;45,60,63;auto;off" """
import hashlib
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from rtamd import scenes
from rtamd.renderer import Camera, MegakernelRenderer, Scene

worlds = [int(v) for v in sys.argv[1].split(",")]
cases = sys.argv[2].split(";")
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
for world in worlds:
    for case in cases:
        os.environ.pop("RT_MEGA_SLICE_BOUNDS", None)
        r = MegakernelRenderer(sc, (W, H), 10, spp); r.set_tile(0, world, 8)
        if case == "off": r.set_schedule(pixel_slices=0)
        elif case != "auto": os.environ["RT_MEGA_SLICE_BOUNDS"] = case
        fr = r.render_frame(cam, want_f32=True, want_u8=False)
        digest = hashlib.sha1(fr.rgba_f32.tobytes()).hexdigest()[:12]
        best = min(r.render_frame(cam, want_f32=False, want_u8=False).device_ms for _ in range(4))
        print(f"1/{world} {case:>24s} -> {fr.pixel_slices} slices {best:8.2f} ms  {fr.rays / best / 1e3:7.0f} Mrays/s  frame {digest}", flush=True)
        r.close()
