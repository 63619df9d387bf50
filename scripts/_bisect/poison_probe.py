import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, R + "/sycl-ray-tracer_amd")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from rtamd import abi, scenes
from rtamd.renderer import Camera, Scene, WavefrontRenderer
# poison the allocator's free blocks: many small and large buffers of 0xFF, then freed
bufs = [torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda") for n in [256, 512, 1024, 4096, 65536, 1 << 20, 16 << 20] * 8]
torch.cuda.synchronize(); del bufs; torch.cuda.empty_cache()
sd = scenes.get_scene("cornell"); gs = Scene(sd, 0); w, h = 80, 64
cam = Camera.for_scene(sd, (w, h))
# renderers that leave garbage in their freed queues / counters
for i in range(4):
    r = WavefrontRenderer(gs, (w, h), 6, 5); r.set_schedule(finish_depth=abi.RT_SCHED_ALL_BOUNCES, stream_lanes=3); r.render_frame(cam); r.close()
for label, kw in (("graph after set_schedule", dict(hip_graph=True)),) * 3:
    r = WavefrontRenderer(gs, (w, h), 6, 5)
    r.set_schedule(**kw)
    for i in range(6):
        fr = r.render_frame(cam)
        print(label, i, fr.rays, flush=True)
    r.close()
