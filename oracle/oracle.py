"""ctypes binding of the CPU oracle (oracle/oracle_rt.cpp). TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the
product package. PARITY UNPINNED: see the header of oracle_rt.cpp.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ORACLE_DIR = Path(__file__).resolve().parent
REPO_DIR = ORACLE_DIR.parent
sys.path.insert(0, str(REPO_DIR / "sycl-ray-tracer_amd"))
from rtamd import abi  # noqa: E402  (struct layouts of the ABI only)

LIB_PATH = ORACLE_DIR / "build" / "liboracle_rt.so"
_lib = None
_P = C.POINTER


def build() -> None:
    subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(LIB_PATH))
        _lib.orc_scene_create.argtypes = [_P(abi.rt_scene_desc), _P(C.c_void_p)]
        _lib.orc_scene_destroy.argtypes = [C.c_void_p]
        _lib.orc_scene_destroy.restype = None
        _lib.orc_camera_init.argtypes = [_P(abi.rt_camera), C.c_int32, C.c_int32, _P(C.c_float), _P(C.c_float), C.c_float]
        _lib.orc_xorshift.argtypes = [C.c_uint32, C.c_uint32, _P(C.c_float), _P(C.c_uint32)]
        _lib.orc_xorshift.restype = None
        _lib.orc_half_roundtrip.argtypes = [C.c_uint32, _P(C.c_float), _P(C.c_float), _P(C.c_uint16)]
        _lib.orc_half_roundtrip.restype = None
        _lib.orc_intersect.argtypes = [C.c_void_p, C.c_uint32, _P(C.c_float), _P(C.c_float), C.c_int, _P(C.c_float),
                                       _P(C.c_float), _P(C.c_float), _P(C.c_uint32)]
        _lib.orc_scatter.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _P(C.c_float), _P(C.c_float), _P(C.c_float),
                                     _P(C.c_uint32), _P(C.c_uint8), _P(C.c_float), _P(C.c_float), _P(C.c_uint32)]
        _lib.orc_render.argtypes = [C.c_void_p, _P(abi.rt_camera), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                    C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_float), _P(C.c_uint8),
                                    _P(C.c_uint64)]
        _lib.orc_hardware_threads.restype = C.c_int
    return _lib


def camera(width: int, height: int, position, direction, focal: float) -> abi.rt_camera:
    cam = abi.rt_camera()
    p = (C.c_float * 3)(*[float(v) for v in position])
    d = (C.c_float * 3)(*[float(v) for v in direction])
    lib().orc_camera_init(C.byref(cam), width, height, p, d, float(focal))
    return cam


def xorshift(seed: int, n: int):
    out = np.zeros(n, np.float32)
    st = C.c_uint32(0)
    lib().orc_xorshift(seed, n, abi.fptr(out), C.byref(st))
    return out, st.value


def half_roundtrip(x: np.ndarray):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    bits = np.zeros(x.shape, np.uint16)
    lib().orc_half_roundtrip(x.size, abi.fptr(x), abi.fptr(out), bits.ctypes.data_as(_P(C.c_uint16)))
    return out, bits


class OracleScene:
    def __init__(self, desc):
        self.desc = desc
        self._c = desc.to_c()
        self.h = C.c_void_p()
        rc = lib().orc_scene_create(C.byref(self._c), C.byref(self.h))
        if rc != 0:
            raise ValueError("oracle: invalid scene description")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def intersect(self, org: np.ndarray, dirs: np.ndarray, use_bvh: bool):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = org.shape[0]
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        tri = np.zeros(n, np.uint32)
        lib().orc_intersect(self.h, n, abi.fptr(org), abi.fptr(dirs), int(use_bvh), abi.fptr(t), abi.fptr(u),
                            abi.fptr(v), abi.u32ptr(tri))
        return t, u, v, tri

    def scatter(self, material: int, dirs, normals, uvs, seeds):
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        uvs = np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        seeds = np.ascontiguousarray(seeds, np.uint32)
        n = dirs.shape[0]
        ok = np.zeros(n, np.uint8)
        od, oa = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        so = np.zeros(n, np.uint32)
        rc = lib().orc_scatter(self.h, material, n, abi.fptr(dirs), abi.fptr(normals), abi.fptr(uvs), abi.u32ptr(seeds),
                               abi.u8ptr(ok), abi.fptr(od), abi.fptr(oa), abi.u32ptr(so))
        assert rc == 0
        return ok, od, oa, so

    def render(self, cam: abi.rt_camera, kind: int, max_depth: int, spp: int, seed_mode: int = 0, use_bvh: bool = True,
               nthreads: int = 0, rank: int = 0, world: int = 1, strip_rows: int = 8, rr_start: int = 0):
        """Returns (rgba_f32 [rows,W,4], rgba_u8 [rows,W,4], rays) for this rank's strips. rr_start > 0 turns on the
        Russian-roulette extension from that bounce on (off in every parity run against the reference's semantics)."""
        W, H = cam.width, cam.height
        rows = sum(1 for y in range(H) if (y // strip_rows) % world == rank)
        f = np.zeros((rows, W, 4), np.float32)
        b = np.zeros((rows, W, 4), np.uint8)
        rays = C.c_uint64(0)
        if nthreads <= 0:
            nthreads = hardware_threads()
        rc = lib().orc_render(self.h, C.byref(cam), kind, seed_mode, max_depth, spp, int(use_bvh), nthreads, rank, world,
                              strip_rows, rr_start, abi.fptr(f), abi.u8ptr(b), C.byref(rays))
        assert rc == 0
        return f, b, rays.value


def hardware_threads() -> int:
    """Host threads this process may really use: min(hardware threads, CPU affinity, cgroup CPU quota)."""
    import os
    n = lib().orc_hardware_threads()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)
