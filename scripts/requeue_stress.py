"""Stress of the dynamic-queue schedule (RT_WF_SAMPLES_PER_LAUNCH=C, RT_WF_REQUEUE=1): the same frame many times at several sizes and
C values; every frame must equal the first one bit for bit (a stale or torn hand-over would change a pixel), and the first one must
equal the default schedule's frame (which the parity tests pin to the oracle)."""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import scenes
from rtamd.renderer import Scene, Camera, WavefrontRenderer
W, H, spp, depth, frames = (int(v) for v in sys.argv[1:6])
sd = scenes.atrium_scene(2); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (W, H))
r = WavefrontRenderer(sc, (W, H), depth, spp)
first = r.render_frame(cam)
bad = 0
for i in range(frames - 1):
    fr = r.render_frame(cam)
    if fr.rays != first.rays or not np.array_equal(fr.rgba_f32, first.rgba_f32): bad += 1
np.save(sys.argv[6], first.rgba_f32)
print(first.rays, bad)
'''
def run(env, args, out):
    p = subprocess.run([sys.executable, "-c", CHILD, *map(str, args), out], cwd=REPO, env=dict(os.environ, **env), capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout.split()
import numpy as np
ok = True
for (w, h, spp, depth, frames) in ((1920, 1080, 16, 10, 12), (1280, 720, 32, 16, 12), (777, 333, 24, 10, 20), (3840, 2160, 8, 6, 4)):
    ref = run({}, (w, h, spp, depth, 1), "/tmp/ref.npy")
    for c in (4, 16, spp):
        got = run({"RT_WF_SAMPLES_PER_LAUNCH": str(c), "RT_WF_REQUEUE": "1"}, (w, h, spp, depth, frames), "/tmp/got.npy")
        same = got[0] == ref[0] and np.array_equal(np.load("/tmp/ref.npy"), np.load("/tmp/got.npy"))
        print(f"{w}x{h} {spp} spp depth {depth}, C={c}: {frames} frames, {got[1]} differ from the first; first == default schedule: {same}", flush=True)
        ok = ok and same and got[1] == "0"
print("requeue stress", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
