"""ctypes view of include/rt_mi355x.h and the loader of librt_mi355x.so.

This is plumbing only: struct layouts and symbol prototypes. The product path has NO CPU
fallback: `load_library()` raises if the HIP library has not been built.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent          # sycl-ray-tracer_amd/
REPO_DIR = PKG_DIR.parent
LIB_PATH = PKG_DIR / "csrc" / "build" / "librt_mi355x.so"

RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_OOM, RT_ERR_UNSUPPORTED = -1, -2, -3, -4, -5

RT_MAT_NONE, RT_MAT_DIFFUSE, RT_MAT_METALLIC, RT_MAT_DIELECTRIC = 0, 1, 2, 3
RT_TEX_COLOR, RT_TEX_IMAGE = 0, 1
RT_RENDERER_MEGAKERNEL, RT_RENDERER_WAVEFRONT = 0, 1
RT_SEED_DEFAULT, RT_SEED_WAVEFRONT, RT_SEED_MEGAKERNEL = 0, 1, 2
RT_BVH_DEFAULT, RT_BVH_LBVH, RT_BVH_SAH, RT_BVH_LBVH_GPU = 0, 1, 2, 3
RT_TEX_SIZE = 512
RT_TEX_MAX_LAYERS = 128


class rt_camera(C.Structure):
    _fields_ = [
        ("center", C.c_float * 3),
        ("pixel00", C.c_float * 3),
        ("delta_u", C.c_float * 3),
        ("delta_v", C.c_float * 3),
        ("width", C.c_int32),
        ("height", C.c_int32),
    ]


class rt_material(C.Structure):
    _fields_ = [
        ("type", C.c_uint32),
        ("tex_kind", C.c_uint32),
        ("color", C.c_float * 3),
        ("tex_layer", C.c_uint32),
        ("emissive", C.c_float * 3),
        ("roughness", C.c_float),
        ("ior", C.c_float),
    ]


class rt_instance(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 16),
        ("normal_mat", C.c_float * 9),
        ("material", C.c_uint32),
    ]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_uint32),
        ("positions", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
        ("uvs", C.POINTER(C.c_float)),
        ("n_triangles", C.c_uint32),
        ("indices", C.POINTER(C.c_uint32)),
        ("tri_instance", C.POINTER(C.c_uint32)),
        ("n_instances", C.c_uint32),
        ("instances", C.POINTER(rt_instance)),
        ("n_materials", C.c_uint32),
        ("materials", C.POINTER(rt_material)),
        ("n_layers", C.c_uint32),
        ("textures", C.POINTER(C.c_uint8)),
        ("sky", C.c_float * 3),
    ]


class rt_scene_info_t(C.Structure):
    _fields_ = [
        ("n_triangles", C.c_uint32),
        ("n_nodes", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("max_leaf_tris", C.c_uint32),
        ("bounds_lo", C.c_float * 3),
        ("bounds_hi", C.c_float * 3),
        ("sah_cost", C.c_double),
        ("device_bytes", C.c_uint64),
        ("n_leaf_records", C.c_uint32),
        ("n_split_triangles", C.c_uint32),
    ]


RT_SCHED_ALL_BOUNCES = 0xFFFFFFFF
# kernel families of rt_stats.launches_by_kernel (include/rt_mi355x.h)
KERNELS = {"megakernel": 0, "wf_init": 1, "wf_generate": 2, "wf_extend": 3, "wf_shade": 4, "wf_shade_reorder": 5, "wf_shade_matsort": 6,
           "wf_finish": 7, "wf_finish_requeue": 8, "wf_tile_order": 9, "wf_resolve": 10, "fill_black": 11, "wf_shoot": 12}
RT_K_COUNT = 16


class rt_schedule(C.Structure):
    _fields_ = [
        ("finish_depth", C.c_uint32),
        ("samples_per_launch", C.c_uint32),
        ("stream_lanes", C.c_uint32),
        ("requeue", C.c_int32),
        ("reorder", C.c_uint32),
        ("matsort", C.c_uint32),
        ("cost_order", C.c_int32),
        ("hip_graph", C.c_uint32),
        ("fused_bounce", C.c_uint32),
        ("pixel_slices", C.c_int32),
    ]


class rt_stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("seconds", C.c_double),
        ("device_ms", C.c_double),
        ("hot_kernel_ms", C.c_double),
        ("hot_kernel_launches", C.c_uint32),
        ("launches", C.c_uint32),
        ("launches_by_kernel", C.c_uint32 * RT_K_COUNT),
        ("stream_lanes", C.c_uint32),
        ("samples_per_launch", C.c_uint32),
        ("finish_depth", C.c_uint32),
        ("cost_ordered", C.c_uint32),
        ("kernel_ms", C.c_double * RT_K_COUNT),
        ("hw_queues", C.c_uint32),
        ("pixel_slices", C.c_uint32),
    ]


assert C.sizeof(rt_material) == 44
assert C.sizeof(rt_instance) == 104
assert C.sizeof(rt_camera) == 56

# name -> (restype, argtypes): every symbol include/rt_mi355x.h declares
_P = C.POINTER
PROTOTYPES = {
    "rt_camera_init": (C.c_int, [_P(rt_camera), C.c_int32, C.c_int32, _P(C.c_float), _P(C.c_float), C.c_float]),
    "rt_scene_create": (C.c_int, [_P(rt_scene_desc), C.c_int, C.c_int, _P(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_scene_info": (C.c_int, [C.c_void_p, _P(rt_scene_info_t)]),
    "rt_scene_check_bvh": (C.c_int, [C.c_void_p]),
    "rt_scene_count_visits": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_float), _P(C.c_float), C.c_int, _P(C.c_uint64), _P(C.c_uint64),
                                         _P(C.c_float), _P(C.c_uint32)]),
    "rt_intersect_batch": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_float), _P(C.c_float), _P(C.c_float),
                                     _P(C.c_float), _P(C.c_float), _P(C.c_uint32)]),
    "rt_renderer_create": (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                     C.c_uint32, _P(C.c_void_p)]),
    "rt_renderer_destroy": (None, [C.c_void_p]),
    "rt_renderer_set_tile": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_renderer_local_rows": (C.c_int32, [C.c_void_p]),
    "rt_renderer_global_row": (C.c_int32, [C.c_void_p, C.c_int32]),
    "rt_renderer_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_renderer_set_russian_roulette": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rt_renderer_set_schedule": (C.c_int, [C.c_void_p, _P(rt_schedule)]),
    "rt_renderer_get_schedule": (C.c_int, [C.c_void_p, _P(rt_schedule)]),
    "rt_render_frame_begin": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_render_frame_end": (C.c_int, [C.c_void_p, C.POINTER(rt_stats)]),
    "rt_render_frame": (C.c_int, [C.c_void_p, _P(rt_camera), _P(C.c_float), _P(C.c_uint8), _P(rt_stats)]),
    "rt_render_frame_device": (C.c_int, [C.c_void_p, _P(rt_camera), C.c_void_p, C.c_void_p, C.c_void_p,
                                         _P(rt_stats)]),
    "rt_comm_create": (C.c_int, [C.c_int, _P(C.c_int), _P(C.c_void_p)]),
    "rt_comm_destroy": (None, [C.c_void_p]),
    "rt_comm_uses_rccl": (C.c_int, [C.c_void_p]),
    "rt_renderer_tile_f32": (C.c_void_p, [C.c_void_p]),
    "rt_renderer_tile_u8": (C.c_void_p, [C.c_void_p]),
    "rt_frame_gather": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_float), _P(C.c_uint8), C.c_int, C.c_int]),
    "rt_frame_gather_begin": (C.c_int, [C.c_void_p, _P(C.c_void_p), C.c_int, C.c_int]),
    "rt_comm_wait": (C.c_int, [C.c_void_p, _P(C.c_float), _P(C.c_uint8)]),
    "rt_comm_size": (C.c_int, [C.c_void_p]),
    "rt_comm_frame_f32": (C.c_void_p, [C.c_void_p]),
    "rt_comm_frame_u8": (C.c_void_p, [C.c_void_p]),
    "rt_probe_xorshift": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, _P(C.c_float), _P(C.c_uint32)]),
    "rt_probe_half_roundtrip": (C.c_int, [C.c_int, C.c_uint32, _P(C.c_float), _P(C.c_float), _P(C.c_uint16)]),
    "rt_probe_scatter": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _P(C.c_float), _P(C.c_float),
                                   _P(C.c_float), _P(C.c_uint32), _P(C.c_uint8), _P(C.c_float),
                                   _P(C.c_float), _P(C.c_uint32)]),
    "rt_last_error": (C.c_char_p, []),
    "rt_probe_rounding": (C.c_int, [C.c_int, _P(C.c_uint64)]),
    "rt_abi_version": (C.c_int, []),
    "rt_device_count": (C.c_int, []),
}

_lib = None


class RtError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"rt_status {status}: {msg}")
        self.status = status


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """Loads librt_mi355x.so (built in-tree by `make -C sycl-ray-tracer_amd/csrc`). No fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else Path(os.environ.get("RT_MI355X_LIB", LIB_PATH))  # env override: A/B builds
    if not p.exists():
        raise FileNotFoundError(
            f"{p} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C {PKG_DIR / 'csrc'}). There is no CPU fallback for the render path.")
    lib = C.CDLL(str(p), mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status: int, lib: C.CDLL | None = None) -> None:
    """`lib`: the library the failing call was made through (its rt_last_error() is per library and thread); default: the product library."""
    if status != RT_OK:
        msg = (lib or load_library()).rt_last_error()
        raise RtError(status, msg.decode() if msg else "")


DEV_LIB_PATH = PKG_DIR / "csrc" / "build" / "librt_mi355x_dev.so"
_dev_lib = None


def load_developer_library() -> C.CDLL:
    """librt_mi355x_dev.so (`make -C sycl-ray-tracer_amd/csrc dev`): the same sources with -DRT_DEVELOPER_KNOBS — the build whose tuning knobs and
    test hooks (RT_WF_*, RT_MEGA_*, RT_BVH_*, RT_INJECT_ALLOC_FAILURE) read the environment. Tests of those hooks and the sweep scripts use it
    (Scene(..., lib=...)); the product library reads GPU_MAX_HW_QUEUES, RT_PROFILE_KERNELS and RT_KERNEL_STATS only."""
    global _dev_lib
    if _dev_lib is None:
        _dev_lib = load_library(DEV_LIB_PATH)
    return _dev_lib


def fptr(a):
    return a.ctypes.data_as(_P(C.c_float))


def u32ptr(a):
    return a.ctypes.data_as(_P(C.c_uint32))


def u8ptr(a):
    return a.ctypes.data_as(_P(C.c_uint8))
