"""BASELINE config 5's tile (atrium 3840x2160, depth 16, tile 0 of 8) at 16 of its 256 spp: wavefront stream lanes x k_wf_finish occupancy."""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
CHILD = r'''
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
sd = scenes.atrium_scene(4); sc = Scene(sd, 0); W, H = 3840, 2160; cam = Camera.for_scene(sd, (W, H))
cls = MegakernelRenderer if os.environ.get("WHICH") == "mega" else WavefrontRenderer
r = cls(sc, (W, H), 16, 16); r.set_tile(0, 8, 8)
r.render_frame(cam, want_f32=False, want_u8=False)
fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
print(f"{fr.device_ms:.1f} ms  {fr.rays / fr.device_ms / 1e3:.0f} Mrays/s")
'''
print("megakernel:", subprocess.run([sys.executable, "-c", CHILD], cwd=REPO, env=dict(os.environ, WHICH="mega"), capture_output=True, text=True).stdout.strip(), flush=True)
for k, occ in ((1, 6), (2, 3), (2, 4), (2, 6), (3, 2), (4, 2)):
    env = dict(os.environ, RT_WF_STREAMS=str(k), RT_WF_FINISH_OCC=str(occ))
    print(f"wavefront streams {k} occ {occ}:", subprocess.run([sys.executable, "-c", CHILD], cwd=REPO, env=env, capture_output=True, text=True).stdout.strip(), flush=True)
