"""The oracle under AddressSanitizer + UBSan (CPU build only: GPU ASan is not available on this pool). Runs a
small render of every renderer/intersector combination in a subprocess with the sanitizer runtime preloaded."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    so = REPO / "oracle" / "build" / "liboracle_rt_asan.so"
    subprocess.run(["make", "-C", str(REPO / "oracle"), "build/liboracle_rt_asan.so"], check=True, capture_output=True)
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not Path(asan).exists():
        pytest.skip("libasan not found")
    code = f"""
import sys
sys.path.insert(0, {str(REPO)!r}); sys.path.insert(0, {str(REPO / 'sycl-ray-tracer_amd')!r})
from oracle import oracle as O
O.LIB_PATH = O.Path({str(so)!r})
from rtamd import scenes
for name in ("cube", "cornell"):
    sd = scenes.get_scene(name)
    sc = O.OracleScene(sd)
    cam = O.camera(24, 17, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    for kind in (0, 1):
        for bvh in (False, True):
            f, b, rays = sc.render(cam, kind, 5, 2, use_bvh=bvh, nthreads=2, rank=0, world=2, strip_rows=8)
            assert rays > 0
print("sanitizer run ok")
"""
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "sanitizer run ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
