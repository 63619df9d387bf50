"""BASELINE config 5's tile (atrium 3840x2160, depth 16, tile 0 of 8) at 16 of its 256 spp: samples per k_wf_finish launch, breadth first
(RT_WF_REQUEUE=1) and depth first (=0), against the megakernel."""
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
r = cls(sc, (W, H), 16, int(os.environ.get("SPP", "16"))); r.set_tile(0, 8, 8)
r.render_frame(cam, want_f32=False, want_u8=False)
fr = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
print(f"{fr.device_ms:.1f} ms  {fr.rays / fr.device_ms / 1e3:.0f} Mrays/s")
'''
def run(**env):
    return subprocess.run([sys.executable, "-c", CHILD], cwd=REPO, env=dict(os.environ, **{k: str(v) for k, v in env.items()}), capture_output=True, text=True).stdout.strip()
print("megakernel:", run(WHICH="mega"), flush=True)
for c, rq in ((1, 1), (4, 1), (16, 1), (4, 0), (16, 0)):
    print(f"wavefront, {c} samples per launch, requeue {rq}:", run(RT_WF_SAMPLES_PER_LAUNCH=c, RT_WF_REQUEUE=rq), flush=True)
