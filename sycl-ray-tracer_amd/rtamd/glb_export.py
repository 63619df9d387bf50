"""Writes a SceneDesc as a binary glTF (GLB) that follows the reference's authoring conventions
(README.md:46-66 of the reference; src/scene.cpp:80-94,188-254): embedded PNG base-colour textures,
KHR_materials_emissive_strength / _ior / _transmission, scene extras `sky_color`, one camera node.

Used to hand the generated scenes (Cornell box, atrium) to the C++ loader and the `raytracer` CLI exactly as
a Blender export of the real assets would be. Every float is written with enough digits to round-trip
float32 exactly."""
from __future__ import annotations

import io
import json
import math
import struct

import numpy as np

from . import abi
from .scenes import SceneDesc


def _f(v) -> float:
    return float(np.float32(v))


def _camera_quaternion(direction) -> list[float]:
    """Rotation taking (0,0,-1) to `direction` with +y as up (the reference derives the view direction from the
    node rotation: src/scene.cpp:116-121)."""
    d = np.asarray(direction, np.float64)
    d = d / np.linalg.norm(d)
    right = np.cross(d, [0.0, 1.0, 0.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, d)
    m = np.stack([right, up, -d], axis=1)  # columns: x, y, z axes of the camera
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    return [float(v) for v in q]


def export_glb(desc: SceneDesc, path, with_camera: bool = True, jpeg_quality: int | None = None, texture_size: int | None = None) -> None:
    """jpeg_quality: embed the textures as JPEG of that quality instead of PNG; texture_size: resample them to that
    edge length first (so that the loader's decoder and texture baker have real work to do)."""
    from PIL import Image

    blob = bytearray()
    views, accessors = [], []

    def add_view(data: bytes, target=None) -> int:
        while len(blob) % 4:
            blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}
        if target:
            v["target"] = target
        blob.extend(data)
        views.append(v)
        return len(views) - 1

    pos = np.ascontiguousarray(desc.positions, np.float32)
    nrm = np.ascontiguousarray(desc.normals, np.float32)
    uvs = np.ascontiguousarray(desc.uvs, np.float32)
    v_pos, v_nrm, v_uv = add_view(pos.tobytes(), 34962), add_view(nrm.tobytes(), 34962), add_view(uvs.tobytes(), 34962)

    meshes, nodes = [], []
    for i in range(desc.transforms.shape[0]):
        tri = desc.indices[desc.tri_instance == i]
        if tri.size == 0:
            continue
        vmin, vmax = int(tri.min()), int(tri.max())
        cnt = vmax - vmin + 1
        a0 = len(accessors)
        p = pos[vmin: vmax + 1]
        accessors.append({"bufferView": v_pos, "byteOffset": vmin * 12, "componentType": 5126, "count": cnt, "type": "VEC3",
                          "min": [float(x) for x in p.min(0)], "max": [float(x) for x in p.max(0)]})
        accessors.append({"bufferView": v_nrm, "byteOffset": vmin * 12, "componentType": 5126, "count": cnt, "type": "VEC3"})
        accessors.append({"bufferView": v_uv, "byteOffset": vmin * 8, "componentType": 5126, "count": cnt, "type": "VEC2"})
        v_idx = add_view((tri - np.uint32(vmin)).astype(np.uint32).tobytes(), 34963)
        accessors.append({"bufferView": v_idx, "componentType": 5125, "count": int(tri.size), "type": "SCALAR"})
        meshes.append({"primitives": [{"attributes": {"POSITION": a0, "NORMAL": a0 + 1, "TEXCOORD_0": a0 + 2},
                                       "indices": a0 + 3, "material": int(desc.inst_material[i])}]})
        nodes.append({"mesh": len(meshes) - 1, "matrix": [_f(v) for v in desc.transforms[i]]})

    images, textures = [], []
    for layer in range(desc.textures.shape[0]):
        buf = io.BytesIO()
        im = Image.fromarray(desc.textures[layer], "RGBA")
        if texture_size:
            im = im.resize((texture_size, texture_size), Image.BICUBIC)
        if jpeg_quality is None:
            im.save(buf, format="PNG")
        else:
            im.convert("RGB").save(buf, format="JPEG", quality=int(jpeg_quality), subsampling=2)
        images.append({"bufferView": add_view(buf.getvalue()), "mimeType": "image/png" if jpeg_quality is None else "image/jpeg"})
        textures.append({"source": layer})

    materials, used_ext = [], set()
    for m in desc.materials:
        g = {"pbrMetallicRoughness": {"baseColorFactor": [_f(m.color[0]), _f(m.color[1]), _f(m.color[2]), 1.0]}}
        pbr = g["pbrMetallicRoughness"]
        if m.type == abi.RT_MAT_DIELECTRIC:
            g["extensions"] = {"KHR_materials_ior": {"ior": _f(m.ior)}, "KHR_materials_transmission": {"transmissionFactor": 1.0}}
            used_ext |= {"KHR_materials_ior", "KHR_materials_transmission"}
            pbr["metallicFactor"], pbr["roughnessFactor"] = 0.0, 0.0
        elif m.type == abi.RT_MAT_METALLIC:
            pbr["metallicFactor"], pbr["roughnessFactor"] = 1.0, _f(m.roughness)
        else:
            pbr["metallicFactor"], pbr["roughnessFactor"] = 0.0, 0.5
        if m.type != abi.RT_MAT_DIELECTRIC:
            if m.tex_layer is not None:
                pbr["baseColorTexture"] = {"index": int(m.tex_layer)}
            if any(float(e) != 0.0 for e in m.emissive):
                # emissive = emissiveFactor * emissiveStrength (src/scene.cpp:198-206); strength 1 keeps it exact
                g["emissiveFactor"] = [_f(e) for e in m.emissive]
                g.setdefault("extensions", {})["KHR_materials_emissive_strength"] = {"emissiveStrength": 1.0}
                used_ext.add("KHR_materials_emissive_strength")
        materials.append(g)

    scene_nodes = list(range(len(nodes)))
    js = {"asset": {"version": "2.0", "generator": "rtamd.glb_export"}, "scene": 0,
          "scenes": [{"name": desc.name, "nodes": scene_nodes, "extras": {"sky_color": [_f(v) for v in desc.sky]}}],
          "nodes": nodes, "meshes": meshes, "materials": materials, "accessors": accessors, "bufferViews": views}
    if images:
        js["images"], js["textures"] = images, textures
    if used_ext:
        js["extensionsUsed"] = sorted(used_ext)
    if with_camera:
        cam = desc.camera
        js["cameras"] = [{"type": "perspective", "perspective": {"yfov": 2.0 * math.atan(1.0 / float(cam.focal_length)),
                                                                  "znear": 0.01, "aspectRatio": 16.0 / 9.0}}]
        # NB: the reference tests `if (camera_node_index)` (src/scene.cpp:109), so a camera at node 0 is ignored:
        # the camera goes last, never first
        nodes.append({"camera": 0, "translation": [float(v) for v in cam.position], "rotation": _camera_quaternion(cam.direction)})
        js["scenes"][0]["nodes"].append(len(nodes) - 1)
    while len(blob) % 4:
        blob.append(0)
    js["buffers"] = [{"byteLength": len(blob)}]
    jb = json.dumps(js, separators=(",", ":")).encode()
    jb += b" " * ((4 - len(jb) % 4) % 4)
    total = 12 + 8 + len(jb) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, total))
        f.write(struct.pack("<II", len(jb), 0x4E4F534A))
        f.write(jb)
        f.write(struct.pack("<II", len(blob), 0x004E4942))
        f.write(bytes(blob))
