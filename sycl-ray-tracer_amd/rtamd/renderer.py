"""Host-side mirror of the reference's renderer plugin surface over the C ABI.

    IRenderer.render_frame(camera, scene)            src/render.hpp:11-18
    MegakernelRenderer(img_size, max_depth, spp)     src/render_megakernel.hpp:13-19
    WavefrontRenderer(img_size, max_depth, spp)      src/render_wavefront.hpp:55-61
    Camera(img_size, center, dir, focal_length)      src/camera.hpp:74-106

All rendering happens in librt_mi355x.so (HIP, gfx950). There is no CPU path here: constructing a
Scene or renderer without the built library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import abi
from .scenes import SceneDesc


class Camera:
    """== raytracer::Camera; the constructor arithmetic runs in rt_camera_init (host C++)."""

    def __init__(self, img_size, center, direction, focal_length: float):
        self.c = abi.rt_camera()
        lib = abi.load_library()
        ce = (C.c_float * 3)(*[float(v) for v in center])
        di = (C.c_float * 3)(*[float(v) for v in direction])
        abi.check(lib.rt_camera_init(C.byref(self.c), int(img_size[0]), int(img_size[1]), ce, di, float(focal_length)))

    @classmethod
    def for_scene(cls, desc: SceneDesc, img_size):
        p = desc.camera
        return cls(img_size, p.position, p.direction, p.focal_length)


class Scene:
    """Device-resident scene: == what raytracer::Scene hands the kernels (RTCScene + GeometryData)."""

    def __init__(self, desc: SceneDesc, device: int = 0, bvh: int = abi.RT_BVH_DEFAULT, lib=None):
        """`lib`: another build of the library (abi.load_developer_library()); renderers of this scene use the same one."""
        self.desc = desc
        self.device = device
        self._lib = lib or abi.load_library()
        self._c = desc.to_c()
        self.h = C.c_void_p()
        abi.check(self._lib.rt_scene_create(C.byref(self._c), device, bvh, C.byref(self.h)), self._lib)

    def info(self) -> abi.rt_scene_info_t:
        out = abi.rt_scene_info_t()
        abi.check(self._lib.rt_scene_info(self.h, C.byref(out)), self._lib)
        return out

    def check_bvh(self) -> None:
        abi.check(self._lib.rt_scene_check_bvh(self.h), self._lib)

    def intersect(self, org: np.ndarray, dirs: np.ndarray):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = org.shape[0]
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        tri = np.zeros(n, np.uint32)
        abi.check(self._lib.rt_intersect_batch(self.h, n, abi.fptr(org), abi.fptr(dirs), abi.fptr(t), abi.fptr(u),
                                               abi.fptr(v), abi.u32ptr(tri)), self._lib)
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
        abi.check(self._lib.rt_probe_scatter(self.h, material, n, abi.fptr(dirs), abi.fptr(normals), abi.fptr(uvs),
                                             abi.u32ptr(seeds), abi.u8ptr(ok), abi.fptr(od), abi.fptr(oa), abi.u32ptr(so)), self._lib)
        return ok, od, oa, so

    def close(self):
        if self.h:
            self._lib.rt_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class Frame:
    rgba_f32: np.ndarray | None  # (rows, W, 4) float32: sqrt(mean rgb), alpha 1 — pre-quantisation
    rgba_u8: np.ndarray | None   # (rows, W, 4) uint8: the reference's RGBA-unorm8 image
    rays: int
    seconds: float
    device_ms: float
    hot_kernel_ms: float
    hot_kernel_launches: int
    launches: int
    kernels: dict = None         # launches of the frame per kernel family (abi.KERNELS): what actually ran
    stream_lanes: int = 0        # the schedule as the library resolved it for this tile
    samples_per_launch: int = 0
    finish_depth: int = 0
    cost_ordered: bool = False
    kernel_ms: dict = None       # with profiling: summed launch durations per kernel family (hipEvents on the launches' own streams)
    hw_queues: int = 0           # GPU_MAX_HW_QUEUES as the library read it (4 = HIP's default): bounds the automatic stream lanes
    pixel_slices: int = 0        # megakernel: slices a pixel's samples were rendered in (1 = every pixel on one lane)

    @classmethod
    def from_stats(cls, f, b, st):
        return cls(f, b, int(st.rays), st.seconds, st.device_ms, st.hot_kernel_ms, int(st.hot_kernel_launches), int(st.launches),
                   {name: int(st.launches_by_kernel[i]) for name, i in abi.KERNELS.items()}, int(st.stream_lanes),
                   int(st.samples_per_launch), int(st.finish_depth), bool(st.cost_ordered),
                   {name: float(st.kernel_ms[i]) for name, i in abi.KERNELS.items()}, int(st.hw_queues), int(st.pixel_slices))

    def stat_lines(self) -> list[str]:
        """The three lines benchmark.py scrapes (src/render_wavefront.cpp:425-427, benchmark.py:49-55)."""
        secs = self.device_ms * 1e-3
        return [f"Time measured: {secs:.6f} seconds", f"Total rays: {self.rays}",
                f"Rays/sec: {self.rays / secs / 1e6:.2f}M"]


class IRenderer:
    """== raytracer::IRenderer (src/render.hpp:11-18)."""
    KIND = -1

    def __init__(self, scene: Scene, img_size, max_depth: int = 10, sample_count: int = 32,
                 seed_mode: int = abi.RT_SEED_DEFAULT):
        self.scene = scene
        self.img_size = (int(img_size[0]), int(img_size[1]))
        self.max_depth, self.sample_count = int(max_depth), int(sample_count)
        self._lib = scene._lib
        self.h = C.c_void_p()
        abi.check(self._lib.rt_renderer_create(self.KIND, scene.h, self.img_size[0], self.img_size[1], self.max_depth,
                                               self.sample_count, seed_mode, C.byref(self.h)), self._lib)

    def set_tile(self, rank: int, world: int, strip_rows: int = 8) -> None:
        abi.check(self._lib.rt_renderer_set_tile(self.h, rank, world, strip_rows), self._lib)

    def set_profiling(self, enable: bool) -> None:
        abi.check(self._lib.rt_renderer_set_profiling(self.h, int(enable)), self._lib)

    def set_russian_roulette(self, start_bounce: int) -> None:
        """Extension (a to-do upstream: PLAN.md:23-27): paths are thinned from bounce `start_bounce` on; 0 = off (default)."""
        abi.check(self._lib.rt_renderer_set_russian_roulette(self.h, int(start_bounce)), self._lib)

    def set_schedule(self, finish_depth: int = 0, samples_per_launch: int = 0, stream_lanes: int = 0, requeue: int = -1,
                     reorder: bool = False, matsort: bool = False, cost_order: int = -1, hip_graph: bool = False,
                     fused_bounce: bool = False, pixel_slices: int = -1) -> None:
        """rt_renderer_set_schedule: which of the wavefront renderer's schedules renders the frame (same frame bit for bit; the
        reference has one: a launch per bounce, src/render_wavefront.cpp:396-417 = finish_depth=abi.RT_SCHED_ALL_BOUNCES).
        Frame.kernels reports what ran. The megakernel uses pixel_slices only (-1 automatic, 0 / 1 off, 2 .. 8: rt_mi355x.h)."""
        sc = abi.rt_schedule(int(finish_depth), int(samples_per_launch), int(stream_lanes), int(requeue), int(bool(reorder)),
                             int(bool(matsort)), int(cost_order), int(bool(hip_graph)), int(bool(fused_bounce)), int(pixel_slices))
        abi.check(self._lib.rt_renderer_set_schedule(self.h, C.byref(sc)), self._lib)

    def get_schedule(self) -> abi.rt_schedule:
        sc = abi.rt_schedule()
        abi.check(self._lib.rt_renderer_get_schedule(self.h, C.byref(sc)), self._lib)
        return sc

    @property
    def local_rows(self) -> int:
        return int(self._lib.rt_renderer_local_rows(self.h))

    def global_rows(self) -> np.ndarray:
        return np.array([self._lib.rt_renderer_global_row(self.h, i) for i in range(self.local_rows)], np.int64)

    def render_frame(self, camera: Camera, scene: Scene | None = None, want_f32: bool = True, want_u8: bool = True) -> Frame:
        if scene is not None and scene is not self.scene:
            raise ValueError("renderer was created for a different scene")
        rows, w = self.local_rows, self.img_size[0]
        f = np.zeros((rows, w, 4), np.float32) if want_f32 else None
        b = np.zeros((rows, w, 4), np.uint8) if want_u8 else None
        st = abi.rt_stats()
        abi.check(self._lib.rt_render_frame(self.h, C.byref(camera.c), abi.fptr(f) if want_f32 else None,
                                            abi.u8ptr(b) if want_u8 else None, C.byref(st)), self._lib)
        return Frame.from_stats(f, b, st)

    def render_frame_device(self, camera: Camera, d_f32: int = 0, d_u8: int = 0, stream: int = 0) -> Frame:
        """Outputs go to DEVICE pointers (e.g. torch tensor .data_ptr()); nothing is copied to host."""
        st = abi.rt_stats()
        abi.check(self._lib.rt_render_frame_device(self.h, C.byref(camera.c), C.c_void_p(d_f32 or None),
                                                   C.c_void_p(d_u8 or None), C.c_void_p(stream or None), C.byref(st)), self._lib)
        return Frame.from_stats(None, None, st)

    def begin_frame(self, camera: Camera, d_f32: int = 0, d_u8: int = 0, stream: int = 0) -> None:
        """Enqueues the frame and returns at once; collect it with end_frame(). Frames of different renderers overlap on
        the device (the next frame's waves move in while this one's last pixels drain)."""
        abi.check(self._lib.rt_render_frame_begin(self.h, C.byref(camera.c), C.c_void_p(d_f32 or None), C.c_void_p(d_u8 or None),
                                                  C.c_void_p(stream or None)), self._lib)

    def end_frame(self) -> Frame:
        st = abi.rt_stats()
        abi.check(self._lib.rt_render_frame_end(self.h, C.byref(st)), self._lib)
        return Frame.from_stats(None, None, st)

    def close(self):
        if self.h:
            self._lib.rt_renderer_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MegakernelRenderer(IRenderer):
    KIND = abi.RT_RENDERER_MEGAKERNEL


class WavefrontRenderer(IRenderer):
    KIND = abi.RT_RENDERER_WAVEFRONT


class TileComm:
    """One process driving several GPUs: the frame's interleaved strips are rendered by one renderer per device and brought to
    the root device by ONE grouped ncclGather over xGMI + a de-interleave kernel (rt_comm_create / rt_frame_gather in the C ABI;
    SURVEY §8e). With the same device listed more than once (a rehearsal of an N-GPU split on fewer GPUs) RCCL cannot be used
    (one rank per device) and the strips move by device copies instead: `uses_rccl` tells which."""

    def __init__(self, devices, lib=None):
        self._lib = lib or abi.load_library()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self.h = C.c_void_p()
        abi.check(self._lib.rt_comm_create(len(self.devices), arr, C.byref(self.h)), self._lib)

    @property
    def uses_rccl(self) -> bool:
        return bool(self._lib.rt_comm_uses_rccl(self.h))

    def render_and_gather(self, renderers, camera: Camera, want_f32: bool = True, want_u8: bool = True):
        """renderers[i] = tile (i, n) on devices[i]. All tiles are enqueued (each into its renderer's own device buffers, no
        host copy), collected, gathered to the root and returned as (full rgba_f32 | None, full rgba_u8 | None, rays)."""
        n = len(self.devices)
        assert len(renderers) == n
        for r in renderers:
            r.begin_frame(camera, d_f32=self._lib.rt_renderer_tile_f32(r.h) if want_f32 else 0,
                          d_u8=self._lib.rt_renderer_tile_u8(r.h) if want_u8 else 0)
        rays = sum(r.end_frame().rays for r in renderers)
        w, h = renderers[0].img_size
        f = np.zeros((h, w, 4), np.float32) if want_f32 else None
        b = np.zeros((h, w, 4), np.uint8) if want_u8 else None
        hs = (C.c_void_p * n)(*[r.h for r in renderers])
        abi.check(self._lib.rt_frame_gather(self.h, hs, abi.fptr(f) if want_f32 else None, abi.u8ptr(b) if want_u8 else None, 0, 0), self._lib)
        return f, b, rays

    @property
    def size(self) -> int:
        return int(self._lib.rt_comm_size(self.h))

    def gather_begin(self, renderers, want_f32: bool = True, want_u8: bool = True) -> None:
        """rt_frame_gather_begin: enqueues the gather of the renderers' collected frames (no host wait); the renderers may begin their
        next frame right away."""
        hs = (C.c_void_p * len(renderers))(*[r.h for r in renderers])
        abi.check(self._lib.rt_frame_gather_begin(self.h, hs, int(want_f32), int(want_u8)), self._lib)

    def wait(self, shape, want_f32: bool = True, want_u8: bool = True):
        """rt_comm_wait: blocks until the gathered frame is complete on the root device and returns host copies of it."""
        h, w = shape
        f = np.zeros((h, w, 4), np.float32) if want_f32 else None
        b = np.zeros((h, w, 4), np.uint8) if want_u8 else None
        abi.check(self._lib.rt_comm_wait(self.h, abi.fptr(f) if want_f32 else None, abi.u8ptr(b) if want_u8 else None), self._lib)
        return f, b

    def close(self):
        if self.h:
            self._lib.rt_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def assemble_tiles(parts: list[np.ndarray], height: int, world: int, strip_rows: int = 8) -> np.ndarray:
    """De-interleaves per-rank compact strip buffers (rank order) into the full frame."""
    w, ch = parts[0].shape[1], parts[0].shape[2]
    out = np.zeros((height, w, ch), parts[0].dtype)
    for rank, p in enumerate(parts):
        rows = [y for y in range(height) if (y // strip_rows) % world == rank]
        out[rows] = p[: len(rows)]
    return out
