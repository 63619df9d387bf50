#!/bin/bash
# scripts/ab_scenes.sh name1 name2 ...: megakernel frame time of kernel variants on both workload classes (atrium 1080p 16 spp, voxel terrain 1080p 32 spp) + parity of
# the variant against the product library on a small frame (bit-equal fp32 frames and ray counts)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  echo "=== $v"
  python3 - <<PY
import sys, hashlib
sys.path.insert(0, "."); sys.path.insert(0, "sycl-ray-tracer_amd")
from rtamd import scenes
from rtamd.renderer import Scene, Camera, MegakernelRenderer, WavefrontRenderer
for name, fn, spp in (("atrium", lambda: scenes.atrium_scene(4), 16), ("voxel", lambda: scenes.voxel_scene(4), 32)):
    sd = fn(); sc = Scene(sd, 0); cam = Camera.for_scene(sd, (1920, 1080))
    for cls in (MegakernelRenderer, WavefrontRenderer):
        r = cls(sc, (1920, 1080), 10, spp)
        r.render_frame(cam, want_f32=False, want_u8=False)
        best = min((r.render_frame(cam, want_f32=False, want_u8=False) for _ in range(3)), key=lambda f: f.device_ms)
        r.close()
        r = cls(sc, (256, 144), 10, 4)
        fr = r.render_frame(cam if False else Camera.for_scene(sd, (256, 144)))
        r.close()
        print(f"   {name:7s} {cls.__name__:18s} {best.device_ms:8.3f} ms {best.rays / best.device_ms / 1e3:8.1f} Mrays/s   small-frame digest {hashlib.md5(fr.rgba_f32.tobytes()).hexdigest()[:12]} rays {fr.rays}", flush=True)
    sc.close()
PY
done
