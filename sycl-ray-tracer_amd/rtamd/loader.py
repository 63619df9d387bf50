"""GLB scenes through the C++ host loader (host/scene_loader.cpp in librt_host.so): node hierarchy, instancing,
materials incl. the emissive-strength / ior / transmission extensions, sky extras, camera node, and embedded PNG / JPEG
textures decoded and baked to 512x512 layers — the restatement of the reference's src/scene.cpp + src/image_manager.hpp.

`rtamd.scenes.load_glb` is the small pure-Python reader used for the two shipped, texture-less assets; this one is what
a user's own GLB (e.g. the real Sponza) goes through, and what the `raytracer` CLI uses."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import abi
from .scenes import CameraPose, Material, SceneDesc

_lib = None


def _host():
    global _lib
    if _lib is None:
        path = Path(__file__).resolve().parent.parent / "host" / "build" / "librt_host.so"
        if not path.exists():
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(str(path))
        lib.rth_last_error.restype = C.c_char_p
        lib.rth_load_glb.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rth_scene_free.argtypes = [C.c_void_p]
        lib.rth_scene_desc.argtypes = [C.c_void_p, C.POINTER(abi.rt_scene_desc)]
        lib.rth_scene_camera.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        _lib = lib
    return _lib


def load_glb(path, name: str | None = None) -> SceneDesc:
    """Loads `path` with the C++ loader and returns an independent SceneDesc (numpy copies of every array)."""
    lib = _host()
    h = C.c_void_p()
    if lib.rth_load_glb(str(path).encode(), C.byref(h)) != 0:
        raise RuntimeError(lib.rth_last_error().decode())
    try:
        d = abi.rt_scene_desc()
        lib.rth_scene_desc(h, C.byref(d))
        arr = lambda p, n, dt: (np.ctypeslib.as_array(p, (n,)).copy() if n else np.zeros(0, dt))
        nv, nt, ni = d.n_vertices, d.n_triangles, d.n_instances
        transforms = np.zeros((ni, 16), np.float32)
        normal_mats = np.zeros((ni, 9), np.float32)
        inst_material = np.zeros(ni, np.uint32)
        for i in range(ni):
            transforms[i] = list(d.instances[i].transform)
            normal_mats[i] = list(d.instances[i].normal_mat)
            inst_material[i] = d.instances[i].material
        materials = []
        for i in range(d.n_materials):
            m = d.materials[i]
            materials.append(Material(int(m.type), tuple(m.color), int(m.tex_layer) if m.tex_kind == abi.RT_TEX_IMAGE else None,
                                      tuple(m.emissive), float(m.roughness), float(m.ior)))
        textures = (np.ctypeslib.as_array(d.textures, (d.n_layers, 512, 512, 4)).copy() if d.n_layers
                    else np.zeros((0, 512, 512, 4), np.uint8))
        pos, dr = (C.c_float * 3)(), (C.c_float * 3)()
        focal, has = C.c_float(), C.c_int()
        lib.rth_scene_camera(h, pos, dr, C.byref(focal), C.byref(has))
        return SceneDesc(name or Path(path).stem, arr(d.positions, 3 * nv, np.float32).reshape(-1, 3), arr(d.normals, 3 * nv, np.float32).reshape(-1, 3),
                         arr(d.uvs, 2 * nv, np.float32).reshape(-1, 2), arr(d.indices, 3 * nt, np.uint32).reshape(-1, 3),
                         arr(d.tri_instance, nt, np.uint32), transforms, normal_mats, inst_material, materials, textures,
                         np.array(list(d.sky), np.float32), CameraPose(tuple(pos), tuple(dr), float(focal.value)))
    finally:
        lib.rth_scene_free(h)
