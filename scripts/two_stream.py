"""Experiment: render the frame as K interleaved sub-tiles on K streams concurrently (hides per-launch tails)."""
import sys, time, threading
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
from rtamd import abi, scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = 1920, 1080
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
for K in (1, 2, 3, 4):
    rs = []
    for k in range(K):
        r = WavefrontRenderer(sc, (W, H), 10, spp); r.set_tile(k, K, 8); rs.append(r)
    def run(r, out, i): out[i] = r.render_frame(cam, want_f32=False, want_u8=False)
    best = 1e9
    for rep in range(3):
        out = [None] * K
        th = [threading.Thread(target=run, args=(rs[i], out, i)) for i in range(K)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        best = min(best, dt)
    rays = sum(o.rays for o in out)
    print(f"K={K} streams: {best*1e3:8.2f} ms  {rays/best/1e6:8.1f} Mrays/s", flush=True)
    for r in rs: r.close()
