"""Frames per second with ONE frame at a time against TWO frames in flight (two renderers, begin/end): how much of the
end-of-frame drain the next frame's waves hide, for the full frame and for the 1/N tiles of the multi-GPU split."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
import torch
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "mega"
W, H, K = 1920, 1080, 8
FS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3]
WORLDS = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 4, 8]
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
cls = MegakernelRenderer if which.startswith("mega") else WavefrontRenderer
for world in WORLDS:
    line = f"{cls.__name__} 1/{world} of the frame:"
    for F in FS:
        rs = [cls(sc, (W, H), 10, spp) for _ in range(F)]
        for r in rs: r.set_tile(0, world, 8); r.render_frame(cam, want_f32=False, want_u8=False)
        bufs = [torch.zeros((rs[0].local_rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(F)]
        torch.cuda.synchronize(); t = time.perf_counter(); q = []; rays = 0
        for i in range(K):
            k = i % F
            if len(q) == F: rays += rs[q.pop(0)].end_frame().rays
            rs[k].begin_frame(cam, d_f32=bufs[k].data_ptr()); q.append(k)
        while q: rays += rs[q.pop(0)].end_frame().rays
        dt = (time.perf_counter() - t) / K
        line += f"  F={F}: {dt*1e3:7.2f} ms/frame ({rays/K/dt/1e6:5.0f} Mrays/s)"
        for r in rs: r.close()
    print(line, flush=True)
