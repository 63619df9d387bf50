"""SHOOT (one kernel per bounce: intersect + shade + compact) under its claim-size knobs, from the environment of the process:
RT_WF_SHOOT_CHUNK (queue slots per cursor claim), RT_WF_SHOOT_TAIL=1 (shares of what is left near the queue's end).
usage: python3 scripts/shoot_probe.py [spp] [lanes ...]   — device time of a frame, Mrays/s; frames are checked against the default schedule's."""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
import numpy as np
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lanes_list = [int(v) for v in sys.argv[2:]] or [1, 2]
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (1920, 1080))
ref = None
if os.environ.get("CHECK"):
    r = WavefrontRenderer(sc, (1920, 1080), 10, spp); ref = r.render_frame(cam, want_u8=False); r.close()
for lanes in lanes_list:
    r = WavefrontRenderer(sc, (1920, 1080), 10, spp)
    r.set_schedule(finish_depth=abi.RT_SCHED_ALL_BOUNCES, fused_bounce=True, stream_lanes=lanes)
    fr = r.render_frame(cam, want_u8=False)
    if ref is not None: assert fr.rays == ref.rays and np.array_equal(fr.rgba_f32, ref.rgba_f32), "frame differs"
    best = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
    e = os.environ.get
    print(f"SHOOT {lanes} lanes chunk={e('RT_WF_SHOOT_CHUNK', 'default')} tail={e('RT_WF_SHOOT_TAIL', '0')} static={e('RT_WF_SHOOT_STATIC_PCT', '-')}: {best.device_ms:9.3f} ms {best.rays / best.device_ms / 1e3:9.1f} Mrays/s", flush=True)
    r.close()
sc.close()
