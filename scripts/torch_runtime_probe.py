"""Why the split per-bounce renderer enqueues more slowly inside bench.py than from a bare process (3 stream lanes, 4,038 launches per frame):
the same frames rendered (a) from a bare process, (b) with torch imported (its wheel's libamdhip64 then serves our library too), (c) as bench.py does:
torch initialised on the GPU, the frame written into a torch tensor, begin_frame / end_frame, profiling off, WALL time over 5 frames.
usage: python3 scripts/torch_runtime_probe.py bare|torch|bench [profiling]"""
import sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "bare"
prof = len(sys.argv) > 2
if mode != "bare":
    import torch
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, WavefrontRenderer
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (1920, 1080))
buf = None
if mode == "bench":
    torch.cuda.set_device(0)
    buf = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
for lanes in (2, 3):
    r = WavefrontRenderer(sc, (1920, 1080), 10, 64); r.set_schedule(finish_depth=abi.RT_SCHED_ALL_BOUNCES, stream_lanes=lanes)
    r.set_tile(0, 1, 8)
    r.set_profiling(prof)
    if mode == "bench":
        r.begin_frame(cam, d_f32=buf.data_ptr()); r.end_frame()
        t0 = time.perf_counter()
        for _ in range(5):
            r.begin_frame(cam, d_f32=buf.data_ptr()); fr = r.end_frame()
        ms = (time.perf_counter() - t0) / 5 * 1e3
    else:
        r.render_frame(cam, want_f32=False, want_u8=False)
        t0 = time.perf_counter()
        for _ in range(5):
            fr = r.render_frame(cam, want_f32=False, want_u8=False)
        ms = (time.perf_counter() - t0) / 5 * 1e3
    print(sys.argv[1:], f"EXTEND+SHADE {lanes} lanes wall {ms:.1f} ms/frame, device {fr.device_ms:.1f} ms, {fr.rays / ms / 1e3:.0f} Mrays/s", flush=True)
    r.close()
