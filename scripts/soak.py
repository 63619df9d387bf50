"""Stability soak on one GPU: repeated scene / renderer creation and destruction (both builders, both renderers, odd sizes,
tile splits), identical results every time, and no device memory drift."""
import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/sycl-ray-tracer_amd')
import numpy as np, torch
from rtamd import scenes, abi
from rtamd.renderer import Scene, Camera, MegakernelRenderer, TileComm, WavefrontRenderer
sd = scenes.atrium_scene(2); t0=time.time()
free0 = None
ref = {}
for it in range(40):
    sc = Scene(sd, 0, abi.RT_BVH_SAH if it % 2 else abi.RT_BVH_LBVH_GPU)
    for cls in (MegakernelRenderer, WavefrontRenderer):
        w, h = (320, 180) if it % 3 else (333, 187)
        r = cls(sc, (w, h), 6, 3)
        if it % 4 == 0: r.set_tile(1, 3, 8)
        fr = r.render_frame(Camera.for_scene(sd, (w, h)))
        key = (cls.__name__, w, it % 4 == 0)
        if key in ref: assert ref[key] == fr.rays and np.array_equal(refimg[key], fr.rgba_u8), key
        else:
            ref[key] = fr.rays; globals().setdefault('refimg', {})[key] = fr.rgba_u8.copy()
        r.close()
    if it % 5 == 0:  # the library's gather: communicator + two tiles created, used and destroyed
        comm = TileComm((0, 0))
        rs = [WavefrontRenderer(sc, (320, 180), 6, 2) for _ in range(2)]
        for k, r in enumerate(rs): r.set_tile(k, 2, 8)
        f, b, rays = comm.render_and_gather(rs, Camera.for_scene(sd, (320, 180)))
        key = ("gather",)
        if key in ref: assert ref[key] == rays and np.array_equal(refimg[key], b), key
        else:
            ref[key] = rays; globals().setdefault('refimg', {})[key] = b.copy()
        for r in rs: r.close()
        comm.close()
    sc.close()
    if it == 7: torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]  # after the runtime's one-time allocations
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"soak ok: 40 scene builds, 80 renderers, {time.time()-t0:.1f} s, device memory drift over the last 32 iterations {(free0-free1)/2**20:.1f} MiB")
