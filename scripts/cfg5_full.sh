#!/bin/bash
# BASELINE config 5 as stated on one GPU's share: atrium 3840x2160, depth 16, 256 spp, tile 0 of 8 - both renderers, default settings
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); W, H = 3840, 2160; cam = Camera.for_scene(sd, (W, H))
for cls in (MegakernelRenderer, WavefrontRenderer):
    r = cls(sc, (W, H), 16, 256); r.set_tile(0, 8, 8)
    r.render_frame(cam, want_f32=False, want_u8=False)
    fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(2)), key=lambda f: f.device_ms)
    print(f"{cls.__name__:20s} {fr.device_ms:8.1f} ms  {fr.rays / fr.device_ms / 1e3:.0f} Mrays/s  rays {fr.rays}", flush=True)
    r.close()
PY
