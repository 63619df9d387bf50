#!/bin/bash
# wavefront schedule: RT_WF_FINISH_DEPTH / RT_WF_SAMPLES_PER_LAUNCH sweeps on the bench workload (16 spp) and on BASELINE config 5's tile
# (4K, depth 16, tile 0 of 8, 16 of its 256 spp)
cd $GRAFT_REPO_ROOT
for c in 1 2 4 8 16; do
  echo "samples_per_launch $c: $(RT_WF_SAMPLES_PER_LAUNCH=$c python3 scripts/quick_bench.py 16 wave 2>&1 | grep Wavefront)"
done
echo "megakernel: $(python3 scripts/quick_bench.py 16 mega 2>&1 | grep Mega)"
python3 - <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); W, H = 3840, 2160; cam = Camera.for_scene(sd, (W, H))
for cls, c in ((MegakernelRenderer, None), (WavefrontRenderer, 1), (WavefrontRenderer, 4), (WavefrontRenderer, 8), (WavefrontRenderer, 16)):
    if c is not None: os.environ["RT_WF_SAMPLES_PER_LAUNCH"] = str(c)
    r = cls(sc, (W, H), 16, 16); r.set_tile(0, 8, 8)
    r.render_frame(cam, want_f32=False, want_u8=False)
    fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(2)), key=lambda f: f.device_ms)
    print(f"config-5 tile (16 of 256 spp) {cls.__name__} samples_per_launch={c}: {fr.device_ms:.1f} ms  {fr.rays / fr.device_ms / 1e3:.0f} Mrays/s", flush=True)
    r.close()
PY
