"""Scene descriptions for the render path: the flat arrays `rt_scene_desc` carries.

Holds (a) `SceneDesc`, the numpy-side owner of those arrays, (b) a small GLB reader that follows
the reference loader's conventions (src/scene.cpp:54-129,164-442,444-510 — TRS*matrix node
transforms, material classification, sky extras, camera node), used for the two shipped assets,
and (c) deterministic procedural generators for the scenes BASELINE.json names but the
reference does not ship (Cornell box, Sponza-like atrium).

Everything here is host-side input preparation; no rendering arithmetic lives in this file.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import struct
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from . import abi

f32 = np.float32


# ----------------------------------------------------------------------------------------------
# small float32 matrix helpers (column-major 4x4 as 16 floats, m[c*4+r]); explicit op order so a
# C++ loader can reproduce them bit for bit
# ----------------------------------------------------------------------------------------------
def mat4_identity() -> np.ndarray:
    m = np.zeros(16, f32)
    m[0] = m[5] = m[10] = m[15] = 1
    return m


def mat4_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros(16, f32)
    for c in range(4):
        for r in range(4):
            acc = f32(a[0 * 4 + r]) * f32(b[c * 4 + 0])
            for k in range(1, 4):
                acc = f32(acc + f32(a[k * 4 + r]) * f32(b[c * 4 + k]))
            out[c * 4 + r] = acc
    return out


def mat4_translate(t) -> np.ndarray:
    m = mat4_identity()
    m[12], m[13], m[14] = f32(t[0]), f32(t[1]), f32(t[2])
    return m


def mat4_scale(s) -> np.ndarray:
    m = mat4_identity()
    m[0], m[5], m[10] = f32(s[0]), f32(s[1]), f32(s[2])
    return m


def mat4_from_quat(q) -> np.ndarray:
    """q = (x, y, z, w) as glTF stores it; glm::mat4_cast arithmetic in float32."""
    x, y, z, w = (f32(v) for v in q)
    one, two = f32(1), f32(2)
    qxx, qyy, qzz = f32(x * x), f32(y * y), f32(z * z)
    qxz, qxy, qyz = f32(x * z), f32(x * y), f32(y * z)
    qwx, qwy, qwz = f32(w * x), f32(w * y), f32(w * z)
    m = mat4_identity()
    m[0] = f32(one - f32(two * f32(qyy + qzz)))
    m[1] = f32(two * f32(qxy + qwz))
    m[2] = f32(two * f32(qxz - qwy))
    m[4] = f32(two * f32(qxy - qwz))
    m[5] = f32(one - f32(two * f32(qxx + qzz)))
    m[6] = f32(two * f32(qyz + qwx))
    m[8] = f32(two * f32(qxz + qwy))
    m[9] = f32(two * f32(qyz - qwx))
    m[10] = f32(one - f32(two * f32(qxx + qyy)))
    return m


def quat_axis_angle(axis, angle: float):
    ax = np.asarray(axis, np.float64)
    ax = ax / np.linalg.norm(ax)
    s = math.sin(angle / 2)
    return (f32(ax[0] * s), f32(ax[1] * s), f32(ax[2] * s), f32(math.cos(angle / 2)))


def trs(t=(0, 0, 0), q=(0, 0, 0, 1), s=(1, 1, 1)) -> np.ndarray:
    """Node::local_matrix (src/scene.cpp:18-21): translate * rotate * scale (* identity matrix)."""
    return mat4_mul(mat4_mul(mat4_translate(t), mat4_from_quat(q)), mat4_scale(s))


def normal_matrix(m4: np.ndarray) -> np.ndarray:
    """GeometryData::obj_to_world = transpose(inverse(mat3(global))) (src/scene.cpp:502), float32,
    column-major 3x3. inverse = adjugate * (1/det), cofactors as a*b - c*d."""
    a = [[f32(m4[c * 4 + r]) for r in range(3)] for c in range(3)]  # a[c][r]
    m00, m01, m02 = a[0]
    m10, m11, m12 = a[1]
    m20, m21, m22 = a[2]
    c00 = f32(f32(m11 * m22) - f32(m21 * m12))
    c01 = f32(f32(m10 * m22) - f32(m12 * m20))
    c02 = f32(f32(m10 * m21) - f32(m11 * m20))
    det = f32(f32(f32(m00 * c00) - f32(m01 * c01)) + f32(m02 * c02))
    inv_det = f32(f32(1) / det)
    inv = np.zeros((3, 3), f32)  # inv[c][r]
    inv[0][0] = f32(c00 * inv_det)
    inv[0][1] = f32(-f32(f32(m01 * m22) - f32(m21 * m02)) * inv_det)
    inv[0][2] = f32(f32(f32(m01 * m12) - f32(m11 * m02)) * inv_det)
    inv[1][0] = f32(-c01 * inv_det)
    inv[1][1] = f32(f32(f32(m00 * m22) - f32(m20 * m02)) * inv_det)
    inv[1][2] = f32(-f32(f32(m00 * m12) - f32(m10 * m02)) * inv_det)
    inv[2][0] = f32(c02 * inv_det)
    inv[2][1] = f32(-f32(f32(m00 * m21) - f32(m20 * m01)) * inv_det)
    inv[2][2] = f32(f32(f32(m00 * m11) - f32(m10 * m01)) * inv_det)
    out = np.zeros(9, f32)
    for c in range(3):
        for r in range(3):
            out[c * 3 + r] = inv[r][c]  # transpose
    return out


# ----------------------------------------------------------------------------------------------
# scene container
# ----------------------------------------------------------------------------------------------
@dataclass
class Material:
    type: int = abi.RT_MAT_DIFFUSE
    color: tuple = (0.8, 0.8, 0.8)
    tex_layer: int | None = None
    emissive: tuple = (0.0, 0.0, 0.0)
    roughness: float = 0.0
    ior: float = 1.5


@dataclass
class CameraPose:
    position: tuple = (0.0, 0.0, 0.0)
    direction: tuple = (0.0, 0.0, -1.0)
    focal_length: float = 1.0


@dataclass
class SceneDesc:
    name: str
    positions: np.ndarray
    normals: np.ndarray
    uvs: np.ndarray
    indices: np.ndarray
    tri_instance: np.ndarray
    transforms: np.ndarray      # (I,16) f32
    normal_mats: np.ndarray     # (I,9)  f32
    inst_material: np.ndarray   # (I,)   u32
    materials: list
    textures: np.ndarray        # (L,512,512,4) u8
    sky: np.ndarray             # (3,) f32
    camera: CameraPose = field(default_factory=CameraPose)
    _keep: list = field(default_factory=list, repr=False)

    @property
    def n_triangles(self) -> int:
        return int(self.indices.shape[0])

    def to_c(self) -> abi.rt_scene_desc:
        """Returns a ctypes rt_scene_desc whose pointers stay valid while `self` is alive."""
        self.positions = np.ascontiguousarray(self.positions, f32)
        self.normals = np.ascontiguousarray(self.normals, f32)
        self.uvs = np.ascontiguousarray(self.uvs, f32)
        self.indices = np.ascontiguousarray(self.indices, np.uint32)
        self.tri_instance = np.ascontiguousarray(self.tri_instance, np.uint32)
        self.textures = np.ascontiguousarray(self.textures, np.uint8)
        n_inst = int(self.transforms.shape[0])
        insts = (abi.rt_instance * max(n_inst, 1))()
        for i in range(n_inst):
            insts[i].transform[:] = [float(v) for v in self.transforms[i]]
            insts[i].normal_mat[:] = [float(v) for v in self.normal_mats[i]]
            insts[i].material = int(self.inst_material[i])
        mats = (abi.rt_material * max(len(self.materials), 1))()
        for i, m in enumerate(self.materials):
            mats[i].type = m.type
            mats[i].tex_kind = abi.RT_TEX_IMAGE if m.tex_layer is not None else abi.RT_TEX_COLOR
            mats[i].color[:] = [float(f32(v)) for v in m.color]
            mats[i].tex_layer = int(m.tex_layer or 0)
            mats[i].emissive[:] = [float(f32(v)) for v in m.emissive]
            mats[i].roughness = float(f32(m.roughness))
            mats[i].ior = float(f32(m.ior))
        d = abi.rt_scene_desc()
        d.n_vertices = self.positions.shape[0]
        d.positions = abi.fptr(self.positions)
        d.normals = abi.fptr(self.normals)
        d.uvs = abi.fptr(self.uvs)
        d.n_triangles = self.indices.shape[0]
        d.indices = abi.u32ptr(self.indices)
        d.tri_instance = abi.u32ptr(self.tri_instance)
        d.n_instances = n_inst
        d.instances = insts
        d.n_materials = len(self.materials)
        d.materials = mats
        d.n_layers = int(self.textures.shape[0])
        d.textures = abi.u8ptr(self.textures) if d.n_layers else None
        d.sky[:] = [float(v) for v in self.sky]
        self._keep = [insts, mats]
        return d

    def world_triangles(self) -> np.ndarray:
        """(T,3,3) float64 world-space vertices (for analytic checks in tests)."""
        out = np.zeros((self.n_triangles, 3, 3))
        for i in range(self.transforms.shape[0]):
            sel = self.tri_instance == i
            m = self.transforms[i].astype(np.float64).reshape(4, 4).T  # row-major
            p = self.positions[self.indices[sel]].astype(np.float64)
            out[sel] = p @ m[:3, :3].T + m[:3, 3]
        return out


class SceneBuilder:
    def __init__(self, name: str):
        self.name = name
        self.pos, self.nrm, self.uv = [], [], []
        self.idx, self.tri_inst = [], []
        self.xf, self.nm, self.imat = [], [], []
        self.materials: list[Material] = []
        self.textures: list[np.ndarray] = []
        self.n_verts = 0
        self.sky = np.array([0.5, 0.7, 1.0], f32)  # Scene::sky_color default (src/scene.hpp:76)
        self.camera = CameraPose()

    def add_material(self, m: Material) -> int:
        self.materials.append(m)
        return len(self.materials) - 1

    def add_texture(self, rgba: np.ndarray) -> int:
        assert rgba.shape == (512, 512, 4) and rgba.dtype == np.uint8
        assert len(self.textures) < abi.RT_TEX_MAX_LAYERS
        self.textures.append(rgba)
        return len(self.textures) - 1

    def add_mesh(self, positions, normals, uvs, indices):
        """Registers vertex data once; returns a handle usable by several instances."""
        positions = np.asarray(positions, f32).reshape(-1, 3)
        normals = np.asarray(normals, f32).reshape(-1, 3)
        uvs = np.asarray(uvs, f32).reshape(-1, 2)
        indices = np.asarray(indices, np.uint32).reshape(-1, 3)
        assert positions.shape[0] == normals.shape[0] == uvs.shape[0]
        assert indices.size == 0 or indices.max() < positions.shape[0]
        base = self.n_verts
        self.pos.append(positions), self.nrm.append(normals), self.uv.append(uvs)
        self.n_verts += positions.shape[0]
        return (base, indices)

    def add_instance(self, mesh, material: int, transform: np.ndarray | None = None) -> int:
        base, indices = mesh
        m = mat4_identity() if transform is None else np.asarray(transform, f32)
        inst = len(self.xf)
        self.xf.append(m), self.nm.append(normal_matrix(m)), self.imat.append(material)
        self.idx.append(indices + np.uint32(base))
        self.tri_inst.append(np.full(indices.shape[0], inst, np.uint32))
        return inst

    def build(self) -> SceneDesc:
        tex = np.stack(self.textures) if self.textures else np.zeros((0, 512, 512, 4), np.uint8)
        return SceneDesc(
            name=self.name,
            positions=np.concatenate(self.pos) if self.pos else np.zeros((0, 3), f32),
            normals=np.concatenate(self.nrm) if self.nrm else np.zeros((0, 3), f32),
            uvs=np.concatenate(self.uv) if self.uv else np.zeros((0, 2), f32),
            indices=np.concatenate(self.idx) if self.idx else np.zeros((0, 3), np.uint32),
            tri_instance=np.concatenate(self.tri_inst) if self.tri_inst else np.zeros((0,), np.uint32),
            transforms=np.stack(self.xf) if self.xf else np.zeros((0, 16), f32),
            normal_mats=np.stack(self.nm) if self.nm else np.zeros((0, 9), f32),
            inst_material=np.asarray(self.imat, np.uint32),
            materials=list(self.materials),
            textures=tex,
            sky=self.sky.astype(f32),
            camera=self.camera,
        )


# ----------------------------------------------------------------------------------------------
# GLB reader (the two shipped assets; PNG/JPEG textures are not decoded here)
# ----------------------------------------------------------------------------------------------
_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def read_glb(path) -> tuple[dict, bytes]:
    b = Path(path).read_bytes()
    magic, version, length = struct.unpack_from("<III", b, 0)
    if magic != 0x46546C67 or version != 2:
        raise ValueError(f"{path}: not a GLB v2 file")
    off, js, bin_chunk = 12, None, b""
    while off < length:
        clen, ctype = struct.unpack_from("<II", b, off)
        data = b[off + 8: off + 8 + clen]
        if ctype == 0x4E4F534A:
            js = json.loads(data)
        elif ctype == 0x004E4942:
            bin_chunk = data
        off += 8 + clen
    return js, bin_chunk


def _accessor(js, blob, i) -> np.ndarray:
    acc = js["accessors"][i]
    view = js["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMP[acc["componentType"]])
    nc = _NCOMP[acc["type"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or dt.itemsize * nc
    out = np.zeros((acc["count"], nc), dt)
    for k in range(acc["count"]):
        out[k] = np.frombuffer(blob, dt, nc, start + k * stride)
    return out


# defaults where the reference's behaviour is undefined
DEFAULT_CAMERAS = {
    "triangle": CameraPose((-3.9, 0.244, 0.218), (1.0, 0.0, 0.0), 1.0),
    "cube": CameraPose((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), 1.0),
}


def camera_direction_of(g: np.ndarray):
    """normalize(quat_cast(global) * (0, 0, -1)) in float32 (src/scene.cpp:113-121): GLM's quat_cast (largest-component form), its
    q * v = v + 2 (w (q x v) + q x (q x v)) and its normalize (v * 1 / sqrt(v . v)); the same arithmetic as host/scene_loader.cpp."""
    M = lambda c, r: f32(g[c * 4 + r])
    fx, fy = f32(f32(M(0, 0) - M(1, 1)) - M(2, 2)), f32(f32(M(1, 1) - M(0, 0)) - M(2, 2))
    fz, fw = f32(f32(M(2, 2) - M(0, 0)) - M(1, 1)), f32(f32(M(0, 0) + M(1, 1)) + M(2, 2))
    biggest, four = 0, fw
    for idx, v in ((1, fx), (2, fy), (3, fz)):
        if v > four:
            biggest, four = idx, v
    big = f32(f32(np.sqrt(f32(four + f32(1)))) * f32(0.5))
    mult = f32(f32(0.25) / big)
    sub = lambda a, b: f32(f32(a - b) * mult)
    add = lambda a, b: f32(f32(a + b) * mult)
    if biggest == 0:
        qw, qx, qy, qz = big, sub(M(1, 2), M(2, 1)), sub(M(2, 0), M(0, 2)), sub(M(0, 1), M(1, 0))
    elif biggest == 1:
        qw, qx, qy, qz = sub(M(1, 2), M(2, 1)), big, add(M(0, 1), M(1, 0)), add(M(2, 0), M(0, 2))
    elif biggest == 2:
        qw, qx, qy, qz = sub(M(2, 0), M(0, 2)), add(M(0, 1), M(1, 0)), big, add(M(1, 2), M(2, 1))
    else:
        qw, qx, qy, qz = sub(M(0, 1), M(1, 0)), add(M(2, 0), M(0, 2)), add(M(1, 2), M(2, 1)), big
    v = (f32(0), f32(0), f32(-1))
    cross = lambda a, b: (f32(f32(a[1] * b[2]) - f32(b[1] * a[2])), f32(f32(a[2] * b[0]) - f32(b[2] * a[0])), f32(f32(a[0] * b[1]) - f32(b[0] * a[1])))
    q3 = (qx, qy, qz)
    uv = cross(q3, v)
    uuv = cross(q3, uv)
    d = [f32(v[k] + f32(f32(f32(uv[k] * qw) + uuv[k]) * f32(2))) for k in range(3)]
    inv = f32(f32(1) / f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])))))
    return tuple(float(f32(c * inv)) for c in d)


def load_glb(path, name: str | None = None) -> SceneDesc:
    js, blob = read_glb(path)
    name = name or Path(path).stem
    sb = SceneBuilder(name)
    # materials: classification of src/scene.cpp:188-254
    gl_mats = js.get("materials", [])
    mat_ids = {}

    def material_for(idx: int) -> int:
        if idx in mat_ids:
            return mat_ids[idx]
        gm = gl_mats[idx] if idx >= 0 else {}  # no material: glTF defaults (tiny_gltf.h:739-743)
        pbr = gm.get("pbrMetallicRoughness", {})
        base = pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])
        metallic = pbr.get("metallicFactor", 1.0)
        rough = pbr.get("roughnessFactor", 1.0)
        ext = gm.get("extensions", {})
        strength = ext.get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 0.0) \
            if "KHR_materials_emissive_strength" in ext else 0.0
        ef = gm.get("emissiveFactor", [0.0, 0.0, 0.0])
        emissive = tuple(f32(f32(e) * f32(strength)) for e in ef)
        tex_layer = None
        if "baseColorTexture" in pbr:
            tex_layer = js["textures"][pbr["baseColorTexture"]["index"]]["source"]
        if "KHR_materials_ior" in ext and "KHR_materials_transmission" in ext:
            m = Material(abi.RT_MAT_DIELECTRIC, ior=ext["KHR_materials_ior"].get("ior", 1.5))
        elif f32(metallic) > f32(0.01):
            m = Material(abi.RT_MAT_METALLIC, tuple(base[:3]), tex_layer, emissive, rough)
        else:
            m = Material(abi.RT_MAT_DIFFUSE, tuple(base[:3]), tex_layer, emissive)
        mat_ids[idx] = sb.add_material(m)
        return mat_ids[idx]

    meshes = []
    for gm in js.get("meshes", []):
        prims = []
        for p in gm["primitives"]:
            a = p["attributes"]
            pos = _accessor(js, blob, a["POSITION"]).astype(f32)
            nrm = _accessor(js, blob, a["NORMAL"]).astype(f32)
            uv = _accessor(js, blob, a["TEXCOORD_0"]).astype(f32)
            idx = _accessor(js, blob, p["indices"]).astype(np.uint32).reshape(-1, 3)
            prims.append((sb.add_mesh(pos, nrm, uv, idx), material_for(p.get("material", -1))))
        meshes.append(prims)

    nodes = js.get("nodes", [])
    scene = js["scenes"][js.get("scene", 0)]
    extras = scene.get("extras", {})
    if isinstance(extras.get("sky_color"), list) and len(extras["sky_color"]) == 3:
        sb.sky = np.array(extras["sky_color"], f32)
    if isinstance(extras.get("sky_strength"), (int, float)):
        sb.sky = (sb.sky * f32(extras["sky_strength"])).astype(f32)

    def local(n) -> np.ndarray:
        """Node::local_matrix (src/scene.cpp:18-21): ((T * R) * S) * matrix, the identity `matrix` of a TRS node included"""
        m = trs(n.get("translation", (0, 0, 0)), n.get("rotation", (0, 0, 0, 1)), n.get("scale", (1, 1, 1)))
        return mat4_mul(m, np.array(n["matrix"], f32) if "matrix" in n else mat4_identity())

    order = []  # instances are attached in NODE INDEX order (src/scene.cpp:101-106)
    cam = None
    locals_, parents = {}, {}

    def global_of(i) -> np.ndarray:
        """Scene::node_global_matrix (src/scene.cpp:137-146), composed from the node UP as the reference does — m = local * scale(global_scale), then
        m = ancestor.local * m for every ancestor, nearest first — not from the root down: fp32 matrix products do not associate."""
        m = mat4_mul(locals_[i], mat4_scale((1.0, 1.0, 1.0)))
        p = parents[i]
        while p is not None:
            m = mat4_mul(locals_[p], m)
            p = parents[p]
        return m

    def visit(i, parent):
        nonlocal cam
        n = nodes[i]
        locals_[i], parents[i] = local(n), parent
        if "camera" in n:
            cam = i
        for c in n.get("children", []):
            visit(c, i)
        if "mesh" in n:
            order.append((i, n["mesh"], global_of(i)))

    for r in scene["nodes"]:
        visit(r, None)
    for _, mesh_i, g in sorted(order, key=lambda t: t[0]):
        for mesh, mat in meshes[mesh_i]:
            sb.add_instance(mesh, mat, g)

    if cam is not None:
        n, g = nodes[cam], global_of(cam)
        yfov = js["cameras"][n["camera"]]["perspective"]["yfov"]
        sb.camera = CameraPose(tuple(float(v) for v in g[12:15]), camera_direction_of(g), float(f32(1.0 / math.tan(yfov / 2.0))))
    elif name in DEFAULT_CAMERAS:
        sb.camera = DEFAULT_CAMERAS[name]
    return sb.build()


# ----------------------------------------------------------------------------------------------
# mesh primitives
# ----------------------------------------------------------------------------------------------
def mesh_quad(p0, p1, p2, p3, nx=1, ny=1, uv_scale=(1.0, 1.0)):
    """Tessellated parallelogram p0 + s*(p1-p0) + t*(p3-p0); normal = (p1-p0) x (p3-p0)."""
    p0, p1, p3 = (np.asarray(v, np.float64) for v in (p0, p1, p3))
    s = np.linspace(0, 1, nx + 1)
    t = np.linspace(0, 1, ny + 1)
    S, T = np.meshgrid(s, t, indexing="xy")
    P = p0 + S[..., None] * (p1 - p0) + T[..., None] * (p3 - p0)
    n = np.cross(p1 - p0, p3 - p0)
    n /= np.linalg.norm(n)
    N = np.broadcast_to(n, P.shape)
    UV = np.stack([S * uv_scale[0], T * uv_scale[1]], -1)
    idx = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i
            b, c, d = a + 1, a + nx + 2, a + nx + 1
            idx += [(a, b, c), (a, c, d)]
    return P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), np.array(idx, np.uint32)


def mesh_box(lo=(-1, -1, -1), hi=(1, 1, 1)):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces = [  # outward normals
        ((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)),  # +x
        ((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),  # -x
        ((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)),  # +y
        ((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)),  # -y
        ((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),  # +z
        ((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)),  # -z
    ]
    P, N, UV, I = [], [], [], []
    for f in faces:
        p, n, uv, idx = mesh_quad(*f)
        I.append(idx + len(P) * 4)
        P.append(p), N.append(n), UV.append(uv)
    return np.concatenate(P), np.concatenate(N), np.concatenate(UV), np.concatenate(I)


def mesh_icosphere(level: int, radius=1.0):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
         (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
         (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    N = np.array(v)
    P = N * radius
    UV = np.stack([0.5 + np.arctan2(N[:, 2], N[:, 0]) / (2 * math.pi), 0.5 - np.arcsin(np.clip(N[:, 1], -1, 1)) / math.pi], -1)
    return P, N, UV, np.array(f, np.uint32)


def mesh_cylinder(segments: int, rings: int, radius=1.0, height=1.0, flute=0.0, flutes=0):
    """Open cylinder around +y from y=0 to y=height; optional fluting (radius modulation)."""
    th = np.linspace(0, 2 * math.pi, segments + 1)
    ys = np.linspace(0, height, rings + 1)
    TH, Y = np.meshgrid(th, ys, indexing="xy")
    R = radius * (1.0 - flute * (0.5 + 0.5 * np.cos(flutes * TH))) if flutes else np.full_like(TH, radius)
    P = np.stack([R * np.cos(TH), Y, R * np.sin(TH)], -1)
    # analytic normal of r(th): (r cos - r' sin, 0, r sin + r' cos) rotated -> outward
    dR = radius * flute * 0.5 * flutes * np.sin(flutes * TH) if flutes else np.zeros_like(TH)
    Nx = R * np.cos(TH) + dR * np.sin(TH)
    Nz = R * np.sin(TH) - dR * np.cos(TH)
    N = np.stack([Nx, np.zeros_like(Nx), Nz], -1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    UV = np.stack([TH / (2 * math.pi) * 2.0, Y / max(height, 1e-9) * 2.0], -1)
    idx = []
    for j in range(rings):
        for i in range(segments):
            a = j * (segments + 1) + i
            b, c, d = a + 1, a + segments + 2, a + segments + 1
            idx += [(a, c, b), (a, d, c)]  # outward winding
    return P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), np.array(idx, np.uint32)


def mesh_torus(seg_major: int, seg_minor: int, R=1.0, r=0.3):
    u = np.linspace(0, 2 * math.pi, seg_major + 1)
    v = np.linspace(0, 2 * math.pi, seg_minor + 1)
    U, V = np.meshgrid(u, v, indexing="xy")
    P = np.stack([(R + r * np.cos(V)) * np.cos(U), r * np.sin(V), (R + r * np.cos(V)) * np.sin(U)], -1)
    N = np.stack([np.cos(V) * np.cos(U), np.sin(V), np.cos(V) * np.sin(U)], -1)
    UV = np.stack([U / (2 * math.pi) * 4, V / (2 * math.pi)], -1)
    idx = []
    for j in range(seg_minor):
        for i in range(seg_major):
            a = j * (seg_major + 1) + i
            b, c, d = a + 1, a + seg_major + 2, a + seg_major + 1
            idx += [(a, c, b), (a, d, c)]
    return P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), np.array(idx, np.uint32)


def mesh_height_sheet(nx: int, ny: int, origin, du, dv, normal, amp: float, waves: float, phase: float):
    """A sheet origin + s*du + t*dv displaced along `normal` by amp*sin(waves*2pi*s + phase)*(0.3+0.7t):
    the curtains. Normals are the analytic surface normals."""
    origin, du, dv, normal = (np.asarray(v, np.float64) for v in (origin, du, dv, normal))
    s = np.linspace(0, 1, nx + 1)
    t = np.linspace(0, 1, ny + 1)
    S, T = np.meshgrid(s, t, indexing="xy")
    w = 2 * math.pi * waves
    Hh = amp * np.sin(w * S + phase) * (0.3 + 0.7 * T)
    dHs = amp * w * np.cos(w * S + phase) * (0.3 + 0.7 * T)
    dHt = amp * np.sin(w * S + phase) * 0.7
    P = origin + S[..., None] * du + T[..., None] * dv + Hh[..., None] * normal
    Ts = du + dHs[..., None] * normal
    Tt = dv + dHt[..., None] * normal
    N = np.cross(Ts, Tt)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    UV = np.stack([S * 3.0, T * 3.0], -1)
    idx = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i
            b, c, d = a + 1, a + nx + 2, a + nx + 1
            idx += [(a, b, c), (a, c, d)]
    return P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), np.array(idx, np.uint32)


def mesh_arch(segments: int, depth_segs: int, radius: float, thickness: float, depth: float):
    """Half-ring (semi-circular arch) in the xy plane extruded along z: inner + outer + two faces."""
    parts = []
    th = np.linspace(0, math.pi, segments + 1)
    zs = np.linspace(-depth / 2, depth / 2, depth_segs + 1)
    for rad, sign in ((radius, -1.0), (radius + thickness, 1.0)):
        TH, Z = np.meshgrid(th, zs, indexing="xy")
        P = np.stack([rad * np.cos(TH), rad * np.sin(TH), Z], -1)
        N = sign * np.stack([np.cos(TH), np.sin(TH), np.zeros_like(TH)], -1)
        UV = np.stack([TH / math.pi * 2, (Z + depth / 2) / depth], -1)
        idx = []
        for j in range(depth_segs):
            for i in range(segments):
                a = j * (segments + 1) + i
                b, c, d = a + 1, a + segments + 2, a + segments + 1
                idx += [(a, b, c), (a, c, d)] if sign > 0 else [(a, c, b), (a, d, c)]
        parts.append((P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), np.array(idx, np.uint32)))
    for z, sign in ((-depth / 2, -1.0), (depth / 2, 1.0)):
        rr = np.array([radius, radius + thickness])
        TH, RR = np.meshgrid(th, rr, indexing="xy")
        P = np.stack([RR * np.cos(TH), RR * np.sin(TH), np.full_like(TH, z)], -1)
        N = np.broadcast_to(np.array([0, 0, sign]), P.shape)
        UV = np.stack([TH / math.pi * 2, (RR - radius) / thickness], -1)
        idx = []
        for i in range(segments):
            a, b, c, d = i, i + 1, i + segments + 2, i + segments + 1
            idx += [(a, c, b), (a, d, c)] if sign > 0 else [(a, b, c), (a, c, d)]
        parts.append((P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), np.array(idx, np.uint32)))
    P, N, UV, I, base = [], [], [], [], 0
    for p, n, uv, idx in parts:
        I.append(idx + base)
        base += p.shape[0]
        P.append(p), N.append(n), UV.append(uv)
    return np.concatenate(P), np.concatenate(N), np.concatenate(UV), np.concatenate(I)


# ----------------------------------------------------------------------------------------------
# procedural 512x512 RGBA8 textures (integer hash noise: no library RNG, stable forever)
# ----------------------------------------------------------------------------------------------
def _hash2(x: np.ndarray, y: np.ndarray, seed: int) -> np.ndarray:
    h = (x.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (y.astype(np.uint32) * np.uint32(0x85EBCA77)) ^ np.uint32(seed)
    h ^= h >> np.uint32(15)
    h = h * np.uint32(0x2C1B3C6D)
    h ^= h >> np.uint32(12)
    h = h * np.uint32(0x297A2D39)
    h ^= h >> np.uint32(15)
    return h


def _value_noise(cells: int, seed: int) -> np.ndarray:
    """Tileable bilinear value noise on a 512x512 grid, values in [0,1]."""
    n = 512
    g = np.arange(n)
    X, Y = np.meshgrid(g, g, indexing="xy")
    fx, fy = X * cells / n, Y * cells / n
    x0, y0 = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
    tx, ty = fx - x0, fy - y0

    def lat(ix, iy):
        return _hash2(ix % cells, iy % cells, seed).astype(np.float64) / 4294967295.0

    a, b = lat(x0, y0), lat(x0 + 1, y0)
    c, d = lat(x0, y0 + 1), lat(x0 + 1, y0 + 1)
    sx, sy = tx * tx * (3 - 2 * tx), ty * ty * (3 - 2 * ty)
    return (a * (1 - sx) + b * sx) * (1 - sy) + (c * (1 - sx) + d * sx) * sy


def _to_rgba(rgb: np.ndarray) -> np.ndarray:
    out = np.zeros((512, 512, 4), np.uint8)
    out[..., :3] = np.clip(np.rint(rgb * 255.0), 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


def texture(kind: str, seed: int = 1, tint=(1.0, 1.0, 1.0)) -> np.ndarray:
    g = np.arange(512)
    X, Y = np.meshgrid(g, g, indexing="xy")
    tint = np.asarray(tint, np.float64)
    noise = 0.5 * _value_noise(8, seed) + 0.3 * _value_noise(32, seed + 1) + 0.2 * _value_noise(128, seed + 2)
    if kind == "checker":
        c = ((X // 64 + Y // 64) % 2).astype(np.float64)
        rgb = (0.25 + 0.6 * c)[..., None] * tint
    elif kind == "tiles":
        edge = ((X % 128 < 4) | (Y % 128 < 4)).astype(np.float64)
        base = 0.55 + 0.25 * noise
        rgb = (base * (1 - 0.6 * edge))[..., None] * tint
    elif kind == "bricks":
        row = Y // 32
        xs = (X + (row % 2) * 32) % 64
        mortar = ((xs < 3) | (Y % 32 < 3)).astype(np.float64)
        brick_id = _hash2((X + (row % 2) * 32) // 64, row, seed).astype(np.float64) / 4294967295.0
        base = 0.45 + 0.2 * brick_id + 0.15 * noise
        rgb = np.stack([base * 1.0, base * 0.62, base * 0.5], -1) * (1 - mortar[..., None]) + \
            mortar[..., None] * np.array([0.7, 0.68, 0.62])
        rgb = rgb * tint
    elif kind == "stone":
        base = 0.6 + 0.3 * (noise - 0.5)
        rgb = np.stack([base, base * 0.97, base * 0.9], -1) * tint
    elif kind == "fabric":
        weave = 0.85 + 0.15 * (((X // 2) + (Y // 2)) % 2)
        stripes = 0.8 + 0.2 * ((X // 32) % 2)
        rgb = (weave * stripes * (0.8 + 0.2 * noise))[..., None] * tint
    elif kind == "marble":
        v = np.sin((X + Y) / 512.0 * 2 * math.pi * 4 + 6.0 * noise) * 0.5 + 0.5
        base = 0.65 + 0.3 * v
        rgb = np.stack([base, base, base * 0.98], -1) * tint
    elif kind == "plaster":
        base = 0.72 + 0.12 * (noise - 0.5)
        rgb = np.stack([base, base * 0.95, base * 0.85], -1) * tint
    elif kind == "wood":
        rings = np.sin((Y / 512.0 * 2 * math.pi * 12) + 4 * noise) * 0.5 + 0.5
        base = 0.35 + 0.25 * rings
        rgb = np.stack([base * 1.0, base * 0.7, base * 0.45], -1) * tint
    else:
        raise ValueError(kind)
    return _to_rgba(np.clip(rgb, 0, 1))


# ----------------------------------------------------------------------------------------------
# scenes
# ----------------------------------------------------------------------------------------------
ASSET_DIR = abi.REPO_DIR / "assets"


def triangle_scene() -> SceneDesc:
    """BASELINE config 1: assets/triangle.glb (1 triangle, diffuse 0.8 grey)."""
    return load_glb(ASSET_DIR / "triangle.glb", "triangle")


def cube_scene() -> SceneDesc:
    """BASELINE config 2: assets/cube.glb (12 triangles, no material -> glTF default -> metallic)."""
    return load_glb(ASSET_DIR / "cube.glb", "cube")


def empty_scene() -> SceneDesc:
    sb = SceneBuilder("empty")
    sb.add_material(Material())
    return sb.build()


def cornell_scene() -> SceneDesc:
    """BASELINE config 3: Cornell box, 116 triangles: 10 wall + 2 emissive light + 12 metallic box +
    12 diffuse box + 80-triangle dielectric icosphere (ior 1.5); black sky; camera yfov 0.6911."""
    sb = SceneBuilder("cornell")
    white = sb.add_material(Material(abi.RT_MAT_DIFFUSE, (0.73, 0.73, 0.73)))
    red = sb.add_material(Material(abi.RT_MAT_DIFFUSE, (0.65, 0.05, 0.05)))
    green = sb.add_material(Material(abi.RT_MAT_DIFFUSE, (0.12, 0.45, 0.15)))
    light = sb.add_material(Material(abi.RT_MAT_DIFFUSE, (0.78, 0.78, 0.78), emissive=(15.0, 15.0, 15.0)))
    metal = sb.add_material(Material(abi.RT_MAT_METALLIC, (0.9, 0.85, 0.7), roughness=0.05))
    glass = sb.add_material(Material(abi.RT_MAT_DIELECTRIC, ior=1.5))
    q = lambda *a, **k: sb.add_mesh(*mesh_quad(*a, **k))
    sb.add_instance(q((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1)), white)     # floor (+y normal)
    sb.add_instance(q((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)), white)         # ceiling (-y)
    sb.add_instance(q((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)), white)     # back (+z)
    sb.add_instance(q((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)), red)       # left (+x)
    sb.add_instance(q((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)), green)         # right (-x)
    sb.add_instance(q((-0.25, 0.995, -0.25), (0.25, 0.995, -0.25), (0.25, 0.995, 0.25), (-0.25, 0.995, 0.25)), light)
    box = sb.add_mesh(*mesh_box())
    sb.add_instance(box, metal, trs((-0.35, -0.4, -0.3), quat_axis_angle((0, 1, 0), 0.3), (0.3, 0.6, 0.3)))
    sb.add_instance(box, white, trs((0.4, -0.7, 0.25), quat_axis_angle((0, 1, 0), -0.35), (0.3, 0.3, 0.3)))
    sph = sb.add_mesh(*mesh_icosphere(1))
    sb.add_instance(sph, glass, trs((0.4, -0.05, 0.25), s=(0.33, 0.33, 0.33)))
    sb.sky = np.zeros(3, f32)
    sb.camera = CameraPose((0.0, 0.0, 3.9), (0.0, 0.0, -1.0), float(f32(1.0 / math.tan(0.6911 / 2.0))))
    return sb.build()


def atrium_scene(detail: int = 4, coarse: bool = False) -> SceneDesc:
    """BASELINE configs 4/5: a deterministic Sponza-like atrium (real Sponza is not available offline).
    Long two-storey hall, colonnades of fluted columns with arches, hanging curtains, open roof,
    8 procedural textures; mostly textured diffuse, two metallic tori, two glass spheres.
    detail=4 gives ~262k triangles; lower values shrink every tessellation for tests.
    coarse=True: floor, walls, gallery slabs and roof are TWO triangles each (the way a modelling tool exports a wall, and the way the real
    Sponza's large polygons are) instead of regular grids — the case spatial splits in the BVH builder exist for; columns, arches,
    curtains and centre pieces stay finely tessellated."""
    d = max(1, int(detail))
    sb = SceneBuilder(f"atrium{d}" + ("c" if coarse else ""))
    T = {k: sb.add_texture(texture(*v)) for k, v in {
        "tiles": ("tiles", 11), "bricks": ("bricks", 23), "stone": ("stone", 37), "marble": ("marble", 41),
        "plaster": ("plaster", 53), "wood": ("wood", 67),
        "fab_r": ("fabric", 71, (0.85, 0.15, 0.12)), "fab_g": ("fabric", 73, (0.15, 0.6, 0.2)),
        "fab_b": ("fabric", 79, (0.15, 0.25, 0.8)),
    }.items()}
    M = {k: sb.add_material(Material(abi.RT_MAT_DIFFUSE, (1, 1, 1), T[k])) for k in T}
    gold = sb.add_material(Material(abi.RT_MAT_METALLIC, (0.95, 0.78, 0.35), roughness=0.2))
    steel = sb.add_material(Material(abi.RT_MAT_METALLIC, (0.8, 0.82, 0.85), T["stone"], roughness=0.2))
    glass = sb.add_material(Material(abi.RT_MAT_DIELECTRIC, ior=1.5))
    L, Wd, Hs = 14.0, 6.0, 4.0  # half length (x), half width (z), storey height
    q = lambda *a, **k: sb.add_mesh(*mesh_quad(*a, **k))
    g = 8 * d
    if coarse:
        g = 0.25  # nx = 4 * g = 1, 2 * g -> 1 (below): one quad per surface
    # ground floor (+y)
    sb.add_instance(q((-L, 0, Wd), (L, 0, Wd), (L, 0, -Wd), (-L, 0, -Wd), nx=max(1, int(4 * g)), ny=max(1, int(2 * g)), uv_scale=(14, 6)), M["tiles"])
    gy = max(1, int(g // 4))
    for sgn in (-1.0, 1.0):
        za, zb = (Wd - 2.0, Wd) if sgn > 0 else (-Wd, -(Wd - 2.0))  # gallery slab spans z in [za, zb]
        yt, yb = Hs, Hs - 0.25
        sb.add_instance(q((-L, yt, zb), (L, yt, zb), (L, yt, za), (-L, yt, za), nx=max(1, int(2 * g)), ny=gy, uv_scale=(14, 1)), M["marble"])   # top, +y
        sb.add_instance(q((-L, yb, za), (L, yb, za), (L, yb, zb), (-L, yb, zb), nx=max(1, int(2 * g)), ny=gy, uv_scale=(14, 1)), M["plaster"])  # underside, -y
        if sgn > 0:  # slab front face and side wall, normals facing the hall (-z)
            sb.add_instance(q((L, yb, za), (-L, yb, za), (-L, yt, za), (L, yt, za), nx=max(1, int(2 * g)), ny=1, uv_scale=(14, 0.1)), M["marble"])
            sb.add_instance(q((L, 0, Wd), (-L, 0, Wd), (-L, 2 * Hs, Wd), (L, 2 * Hs, Wd), nx=max(1, int(4 * g)), ny=max(1, int(g)), uv_scale=(14, 4)), M["bricks"])
        else:        # normals +z
            sb.add_instance(q((-L, yb, zb), (L, yb, zb), (L, yt, zb), (-L, yt, zb), nx=max(1, int(2 * g)), ny=1, uv_scale=(14, 0.1)), M["marble"])
            sb.add_instance(q((-L, 0, -Wd), (L, 0, -Wd), (L, 2 * Hs, -Wd), (-L, 2 * Hs, -Wd), nx=max(1, int(4 * g)), ny=max(1, int(g)), uv_scale=(14, 4)), M["bricks"])
    # end walls, normals facing the hall
    sb.add_instance(q((-L, 0, Wd), (-L, 0, -Wd), (-L, 2 * Hs, -Wd), (-L, 2 * Hs, Wd), nx=max(1, int(2 * g)), ny=max(1, int(g)), uv_scale=(6, 4)), M["plaster"])
    sb.add_instance(q((L, 0, -Wd), (L, 0, Wd), (L, 2 * Hs, Wd), (L, 2 * Hs, -Wd), nx=max(1, int(2 * g)), ny=max(1, int(g)), uv_scale=(6, 4)), M["plaster"])
    # roof slabs (normals -y) leaving a central skylight slot of width 3
    yr = 2 * Hs
    for za, zb in ((-Wd, -1.5), (1.5, Wd)):
        sb.add_instance(q((-L, yr, za), (L, yr, za), (L, yr, zb), (-L, yr, zb), nx=max(1, int(2 * g)), ny=gy, uv_scale=(14, 2)), M["wood"])
    # colonnades: one shared column mesh, instanced
    col = sb.add_mesh(*mesh_cylinder(12 * d, 8 * d, radius=0.32, height=Hs - 0.6, flute=0.12, flutes=12))
    cap = sb.add_mesh(*mesh_box((-0.45, 0, -0.45), (0.45, 0.3, 0.45)))
    ncol = 10
    xs = np.linspace(-L + 1.4, L - 1.4, ncol)
    arch = sb.add_mesh(*mesh_arch(8 * d, max(1, d), radius=(xs[1] - xs[0]) / 2 - 0.45, thickness=0.35, depth=0.7))
    for storey in (0, 1):
        y0 = storey * Hs
        for sgn in (-1.0, 1.0):
            z = sgn * (Wd - 2.0)
            for i, x in enumerate(xs):
                sb.add_instance(cap, M["stone"], trs((x, y0, z)))
                sb.add_instance(col, M["stone"], trs((x, y0 + 0.3, z)))
                sb.add_instance(cap, M["stone"], trs((x, y0 + Hs - 0.3, z)))
                if i + 1 < ncol and storey == 0:
                    sb.add_instance(arch, M["plaster"], trs(((x + xs[i + 1]) / 2, y0 + Hs - 1.3, z)))
    # curtains hanging between upper columns
    fabs = ["fab_r", "fab_g", "fab_b"]
    k = 0
    for sgn in (-1.0, 1.0):
        z = sgn * (Wd - 2.0)
        for i in range(0, ncol - 1, 2):
            x0, x1 = xs[i] + 0.4, xs[i + 1] - 0.4
            mesh = sb.add_mesh(*mesh_height_sheet(16 * d, 16 * d, (x0, 2 * Hs - 0.4, z), (x1 - x0, 0, 0), (0, -(Hs - 1.0), 0),
                                                  (0, 0, -sgn), 0.18, 5.0, 0.7 * k))
            sb.add_instance(mesh, M[fabs[k % 3]])
            k += 1
    # centre pieces
    torus = sb.add_mesh(*mesh_torus(16 * d, 8 * d, R=0.9, r=0.28))
    sb.add_instance(torus, gold, trs((-4.0, 1.2, 0.0), quat_axis_angle((1, 0, 0.3), 1.1)))
    sb.add_instance(torus, steel, trs((5.0, 1.0, 0.5), quat_axis_angle((0.2, 0, 1), 0.8), (0.8, 0.8, 0.8)))
    sph = sb.add_mesh(*mesh_icosphere(min(d, 4)))
    sb.add_instance(sph, glass, trs((0.5, 0.8, -0.6), s=(0.8, 0.8, 0.8)))
    sb.add_instance(sph, glass, trs((-8.0, 0.55, 1.2), s=(0.55, 0.55, 0.55)))
    pedestal = sb.add_mesh(*mesh_box((-0.5, 0, -0.5), (0.5, 0.25, 0.5)))
    sb.add_instance(pedestal, M["wood"], trs((5.0, 0.0, 0.5)))
    sb.sky = np.array([0.5, 0.7, 1.0], f32) * f32(2.0)  # sky_color x sky_strength extras (src/scene.cpp:80-94)
    sb.camera = CameraPose((-12.5, 2.2, 0.4), (1.0, 0.08, -0.03), float(f32(1.0 / math.tan(0.95 / 2.0))))
    return sb.build()


def _block_texture(kind: str, seed: int) -> np.ndarray:
    """Blocky 16x16-texel block faces (each texel 32x32 pixels of the 512^2 layer), as a voxel game's texture atlas entries."""
    g = np.arange(512) // 32
    X, Y = np.meshgrid(g, g, indexing="xy")
    n = _hash2(X, Y, seed).astype(np.float64) / 4294967295.0
    if kind == "grass":
        rgb = np.stack([0.25 + 0.15 * n, 0.55 + 0.25 * n, 0.18 + 0.1 * n], -1)
    elif kind == "dirt":
        rgb = np.stack([0.42 + 0.16 * n, 0.29 + 0.12 * n, 0.18 + 0.08 * n], -1)
        rgb[Y < 3] = np.stack([0.25 + 0.15 * n, 0.55 + 0.25 * n, 0.18 + 0.1 * n], -1)[Y < 3]  # grass fringe at the top of a side face
    elif kind == "stone":
        v = 0.45 + 0.25 * n
        rgb = np.stack([v, v, v * 1.02], -1)
    elif kind == "sand":
        rgb = np.stack([0.82 + 0.1 * n, 0.76 + 0.1 * n, 0.52 + 0.1 * n], -1)
    elif kind == "wood":
        rgb = np.stack([0.40 + 0.1 * n, 0.28 + 0.08 * n, 0.14 + 0.05 * n], -1) * (0.8 + 0.2 * ((X // 2) % 2))[..., None]
    elif kind == "leaves":
        rgb = np.stack([0.12 + 0.1 * n, 0.38 + 0.3 * n, 0.10 + 0.08 * n], -1)
    else:
        raise ValueError(kind)
    return _to_rgba(np.clip(rgb, 0, 1))


def voxel_scene(detail: int = 4) -> SceneDesc:
    """The reference's SECOND workload class (benchmark.py:18 `minecraft.glb`, benchmark_avg.csv:20-21: its rate halves there; the asset is
    git-ignored upstream): a deterministic voxel terrain in the style of a block game. Unit cubes on a (64 x detail)^2 grid — rolling
    hills from three octaves of value noise, a sand-rimmed water table replaced by glass blocks, stone below a dirt crust, trees of trunk
    and leaf blocks, a sprinkling of glowing blocks (emissive) and metal blocks — meshed the way such games do: ONLY faces between a solid
    and an empty cell, one quad (two triangles) per block face, 16 x 16-texel textures sampled nearest. Mostly open sky: short paths
    (the reference's Minecraft scene averages 3.3 rays per path at depth 10 against Sponza's 9.2), many axis-aligned coplanar faces,
    and a tree that is as deep in empty space as in geometry: a different regime from the atrium for the BVH builder and the leaf policy."""
    d = max(1, int(detail))
    n, hmax = 64 * d, 48
    gx, gz = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")

    def noise(cells, seed):  # bilinear value noise over the n x n columns
        fx, fz = gx * cells / n, gz * cells / n
        x0, z0 = np.floor(fx).astype(np.int64), np.floor(fz).astype(np.int64)
        tx, tz = fx - x0, fz - z0
        lat = lambda ix, iz: _hash2(ix, iz, seed).astype(np.float64) / 4294967295.0
        sx, sz = tx * tx * (3 - 2 * tx), tz * tz * (3 - 2 * tz)
        return (lat(x0, z0) * (1 - sx) + lat(x0 + 1, z0) * sx) * (1 - sz) + (lat(x0, z0 + 1) * (1 - sx) + lat(x0 + 1, z0 + 1) * sx) * sz

    h = np.floor(6 + 26 * (0.6 * noise(3, 101) + 0.3 * noise(9, 102) + 0.1 * noise(27, 103))).astype(np.int64)  # column heights
    water = 14
    AIR, GRASS, DIRT, STONE, SAND, GLASS, WOOD, LEAVES, GLOW, METAL = range(10)
    vox = np.zeros((n, hmax, n), np.uint8)
    yy = np.arange(hmax)[None, :, None]
    hh = h[:, None, :]
    vox[(yy < hh - 3)] = STONE
    vox[(yy >= hh - 3) & (yy < hh - 1)] = DIRT
    top = np.where(hh - 1 <= water, SAND, GRASS).astype(np.uint8)
    vox = np.where(yy == hh - 1, top, vox).astype(np.uint8)
    vox[(yy >= hh) & (yy <= water) & np.broadcast_to(hh <= water, vox.shape)] = GLASS  # the lake: dielectric blocks up to the water table
    r = _hash2(gx, gz, 777).astype(np.float64) / 4294967295.0
    ground = (h - 1 > water)
    for (x, z) in zip(*np.nonzero(ground & (r < 0.004) & (gx > 2) & (gx < n - 3) & (gz > 2) & (gz < n - 3))):  # trees
        y0 = int(h[x, z])
        th = 4 + int(_hash2(np.array([x]), np.array([z]), 5)[0] % 3)
        if y0 + th + 3 >= hmax:
            continue
        vox[x, y0:y0 + th, z] = WOOD
        for dy in range(th - 2, th + 2):
            rad = 2 if dy < th else 1
            sub = vox[x - rad:x + rad + 1, y0 + dy, z - rad:z + rad + 1]
            sub[sub == AIR] = LEAVES
    place = lambda lo, hi, kind: vox.__setitem__((gx[ground & (r >= lo) & (r < hi)], h[ground & (r >= lo) & (r < hi)], gz[ground & (r >= lo) & (r < hi)]), kind)
    place(0.0100, 0.0118, GLOW)   # lanterns standing on the ground
    place(0.0200, 0.0212, METAL)

    sb = SceneBuilder(f"voxel{d}")
    T = {k: sb.add_texture(_block_texture(k, 900 + i)) for i, k in enumerate(("grass", "dirt", "stone", "sand", "wood", "leaves"))}
    mats = {GRASS: sb.add_material(Material(abi.RT_MAT_DIFFUSE, (1, 1, 1), T["grass"])), DIRT: sb.add_material(Material(abi.RT_MAT_DIFFUSE, (1, 1, 1), T["dirt"])),
            STONE: sb.add_material(Material(abi.RT_MAT_DIFFUSE, (1, 1, 1), T["stone"])), SAND: sb.add_material(Material(abi.RT_MAT_DIFFUSE, (1, 1, 1), T["sand"])),
            GLASS: sb.add_material(Material(abi.RT_MAT_DIELECTRIC, ior=1.33)), WOOD: sb.add_material(Material(abi.RT_MAT_DIFFUSE, (1, 1, 1), T["wood"])),
            LEAVES: sb.add_material(Material(abi.RT_MAT_DIFFUSE, (1, 1, 1), T["leaves"])),
            GLOW: sb.add_material(Material(abi.RT_MAT_DIFFUSE, (1.0, 0.85, 0.5), emissive=(6.0, 4.5, 2.0))),
            METAL: sb.add_material(Material(abi.RT_MAT_METALLIC, (0.85, 0.86, 0.9), roughness=0.1))}
    # faces: (axis, sign) -> the quad's corner offsets in the order that makes (p1-p0) x (p3-p0) the outward normal
    dirs = [((1, 0, 0), [(1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)]), ((-1, 0, 0), [(0, 0, 1), (0, 1, 1), (0, 1, 0), (0, 0, 0)]),
            ((0, 1, 0), [(0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0)]), ((0, -1, 0), [(0, 0, 1), (0, 0, 0), (1, 0, 0), (1, 0, 1)]),
            ((0, 0, 1), [(1, 0, 1), (1, 1, 1), (0, 1, 1), (0, 0, 1)]), ((0, 0, -1), [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0)])]
    pad = np.zeros((n + 2, hmax + 2, n + 2), np.uint8)
    pad[1:-1, 1:-1, 1:-1] = vox
    pad[:, 0, :] = STONE  # nothing is meshed below the world
    solid = pad != AIR
    per_mat = {k: ([], [], []) for k in mats}
    for nrm, corners in dirs:
        nb = solid[1 + nrm[0]:n + 1 + nrm[0], 1 + nrm[1]:hmax + 1 + nrm[1], 1 + nrm[2]:n + 1 + nrm[2]]
        nb_glass = pad[1 + nrm[0]:n + 1 + nrm[0], 1 + nrm[1]:hmax + 1 + nrm[1], 1 + nrm[2]:n + 1 + nrm[2]] == GLASS
        for kind in mats:
            # a face where the neighbour cell is empty; opaque blocks also show through glass, glass only against air
            show = (vox == kind) & (~nb | (nb_glass & (kind != GLASS)))
            if nrm[1] == 0 and kind == GRASS:
                pass  # (grass blocks show the dirt texture on their sides: handled by assigning side faces of GRASS to DIRT below)
            xs, ys, zs = np.nonzero(show)
            if xs.size == 0:
                continue
            base = np.stack([xs, ys, zs], -1).astype(np.float64)
            quad = base[:, None, :] + np.asarray(corners, np.float64)[None, :, :]          # (F, 4, 3)
            uv = np.broadcast_to(np.array([[0, 1], [0, 0], [1, 0], [1, 1]], np.float64) if nrm[1] == 0 else np.array([[0, 0], [0, 1], [1, 1], [1, 0]], np.float64), (xs.size, 4, 2))
            tgt = DIRT if (kind == GRASS and nrm[1] == 0) else kind
            per_mat[tgt][0].append(quad.reshape(-1, 3)), per_mat[tgt][1].append(np.broadcast_to(np.asarray(nrm, np.float64), (xs.size * 4, 3))), per_mat[tgt][2].append(uv.reshape(-1, 2))
    origin = np.array([n / 2.0, 0.0, n / 2.0])
    for kind, (P, N, UV) in per_mat.items():
        if not P:
            continue
        P, N, UV = np.concatenate(P) - origin, np.concatenate(N), np.concatenate(UV)
        f = np.arange(P.shape[0] // 4, dtype=np.uint32)[:, None] * 4
        idx = np.concatenate([f + np.array([0, 1, 2], np.uint32), f + np.array([0, 2, 3], np.uint32)], 1).reshape(-1, 3)
        sb.add_instance(sb.add_mesh(P, N, UV, idx), mats[kind])
    sb.sky = np.array([0.5, 0.7, 1.0], f32)  # Scene::sky_color default (src/scene.hpp:76)
    eye_y = float(h[n // 8, n // 2] + 9)
    sb.camera = CameraPose((-n / 2.0 + n / 8.0, eye_y, 0.5), (1.0, -0.28, 0.06), float(f32(1.0 / math.tan(1.0 / 2.0))))
    return sb.build()


def rotate_scene(sd: SceneDesc, q, name_suffix="r") -> SceneDesc:
    """The same scene turned as a whole by the unit quaternion q (every instance transform and normal matrix, the camera): what is
    axis-aligned in the original runs diagonally through space — large triangles then have fat bounding boxes (the BVH builder's
    pre-splitting pass exists for those)."""
    import copy
    out = copy.copy(sd)
    R = mat4_from_quat(q)
    out.transforms = np.stack([mat4_mul(R, m) for m in sd.transforms]).astype(f32)
    out.normal_mats = np.stack([normal_matrix(m) for m in out.transforms]).astype(f32)
    r3 = R.reshape(4, 4).T[:3, :3].astype(np.float64)  # column-major storage -> row-major 3x3
    out.camera = CameraPose(tuple(float(v) for v in r3 @ np.asarray(sd.camera.position, np.float64)),
                            tuple(float(v) for v in r3 @ np.asarray(sd.camera.direction, np.float64)), sd.camera.focal_length)
    out.name = sd.name + name_suffix
    out._keep = []
    return out


def atrium_tilted_scene(detail: int = 4, coarse: bool = True) -> SceneDesc:
    """The atrium with two-triangle walls, floor, slabs and roof, turned 33 degrees about (0.3, 1, 0.2): no large surface is axis-aligned."""
    return rotate_scene(atrium_scene(detail, coarse=coarse), quat_axis_angle((0.3, 1.0, 0.2), 0.58))


SCENES = {
    "voxel": voxel_scene,
    "atrium_tilted": atrium_tilted_scene,
    "triangle": triangle_scene,
    "cube": cube_scene,
    "cornell": cornell_scene,
    "atrium": atrium_scene,
    "empty": empty_scene,
}


def get_scene(name: str, **kw) -> SceneDesc:
    return SCENES[name](**kw)


# ----------------------------------------------------------------------------------------------
# a generated scene as ONE file: rank 0 of a multi-process run generates it, the other ranks read it (bench.py)
# ----------------------------------------------------------------------------------------------
def save_scene(sd: SceneDesc, path) -> None:
    """Every array of the description plus the materials / camera as small tables, uncompressed (9 MB of textures: writing is what
    the other ranks wait for)."""
    mats = sd.materials
    np.savez(path, name=np.array(sd.name), positions=sd.positions, normals=sd.normals, uvs=sd.uvs, indices=sd.indices,
             tri_instance=sd.tri_instance, transforms=sd.transforms, normal_mats=sd.normal_mats, inst_material=sd.inst_material,
             textures=sd.textures, sky=sd.sky,
             mat_type=np.array([m.type for m in mats], np.int64), mat_color=np.array([m.color for m in mats], np.float64).reshape(-1, 3),
             mat_tex=np.array([-1 if m.tex_layer is None else m.tex_layer for m in mats], np.int64),
             mat_emissive=np.array([m.emissive for m in mats], np.float64).reshape(-1, 3),
             mat_rough_ior=np.array([(m.roughness, m.ior) for m in mats], np.float64).reshape(-1, 2),
             cam=np.array([*sd.camera.position, *sd.camera.direction, sd.camera.focal_length], np.float64))


def load_scene(path) -> SceneDesc:
    z = np.load(path, allow_pickle=False)
    mats = [Material(int(t), tuple(float(v) for v in c), None if int(x) < 0 else int(x), tuple(float(v) for v in e), float(ri[0]), float(ri[1]))
            for t, c, x, e, ri in zip(z["mat_type"], z["mat_color"], z["mat_tex"], z["mat_emissive"], z["mat_rough_ior"])]
    cam = z["cam"]
    return SceneDesc(str(z["name"]), z["positions"], z["normals"], z["uvs"], z["indices"], z["tri_instance"], z["transforms"], z["normal_mats"],
                     z["inst_material"], mats, z["textures"], z["sky"],
                     CameraPose(tuple(float(v) for v in cam[0:3]), tuple(float(v) for v in cam[3:6]), float(cam[6])))
