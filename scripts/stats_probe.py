import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (1920, 1080))
ALL = abi.RT_SCHED_ALL_BOUNCES
for name, sched in (("finish per sample", dict(samples_per_launch=1, stream_lanes=1)), ("SHOOT 1 lane", dict(finish_depth=ALL, fused_bounce=True, stream_lanes=1)),
                    ("EXTEND+SHADE 1 lane", dict(finish_depth=ALL, stream_lanes=1))):
    r = WavefrontRenderer(sc, (1920, 1080), 10, 4); r.set_schedule(**sched)
    r.render_frame(cam, want_f32=False, want_u8=False)
    print("==", name, flush=True); sys.stderr.flush()
    fr = r.render_frame(cam, want_f32=False, want_u8=False)
    sys.stderr.flush()
    print(f"   {fr.device_ms:.2f} ms {fr.rays/fr.device_ms/1e3:.0f} Mrays/s", flush=True)
    r.close()
