// rt_types.h — data layouts of the device-resident scene and of the kernel parameter block.
// Shared by the host builder (scene_build.cpp) and the HIP kernels (rt_kernels.h, rt_device.h).
#pragma once
#include <cstdint>

namespace rt {

// ---- BVH ----------------------------------------------------------------------------------------
// 4-wide BVH, 64-byte nodes, child boxes quantised to 8 bits per plane on a per-node grid:
//     plane = origin[axis] + q * scale[axis]                  (q = 0..255, scale a power of two)
// quantised conservatively (lo rounded down, hi rounded up) after padding, so a decoded box always
// contains the padded exact box. One node = 4 x dwordx4 loads and decides FOUR children (the first
// version, a BVH2 with fp32 boxes, needed the same 4 loads for two): half the L1 accesses per box
// and half the dependent fetches per ray.
//   word 0-2  origin.xyz (f32)      word 3     scale.x (f32, a power of two: the grid step of the x planes)
//   word 4-9  qlo_x, qhi_x, qlo_y, qhi_y, qlo_z, qhi_z : byte k belongs to child k
//   word 10-11 scale.y, scale.z     word 12-15 child[4]
// (Round 1 packed the three biased exponents into word 3; rebuilding the floats cost 3 x (v_lshlrev 4 + v_and 2 cycles) per
//  node visit on a VALU-bound kernel — profiles/r02_valu_calibration.json — and the two spare words were there.)
// child >= 0: index of an inner node.  child < 0: leaf, ~child = (first << 2) | (count - 1), with
// `first` indexing the leaf-ordered triangle array and 1 <= count <= 4.  child == kChildEmpty: absent.
struct alignas(64) BvhNode {
    float origin[3];
    float scale_x;
    uint32_t q[6]; // qlo_x, qhi_x, qlo_y, qhi_y, qlo_z, qhi_z
    float scale_y, scale_z;
    int32_t child[4];
    float scale(int a) const { const float s = a == 0 ? scale_x : (a == 1 ? scale_y : scale_z); return s < 0.0f ? -s : s; }
    void set_scale(int a, float s) { (a == 0 ? scale_x : (a == 1 ? scale_y : scale_z)) = s; }
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");
constexpr int32_t kChildEmpty = (int32_t)0x80000000;

// In the DEVICE copy of the node array a child word >= 0 is the child node's BYTE offset (index * 64), so that the inner step
// addresses the node with no arithmetic of its own — a 32-bit offset on the scalar base for global memory (the index form cost a
// 64-bit shift and a 64-bit add per step), one right shift for the LDS copy of the tree's top. The host array keeps indices.
constexpr int kMaxLeafTris = 2;    // 1..4 (the leaf code keeps count - 1 in two bits). 4 -> 2: +1.1 % on the atrium — a whole-leaf step runs as many tests as the wave's LARGEST leaf
constexpr int kTopNodes = 341;    // nodes at the head of the node array laid out top first — by surface area: scene_build.cpp; by level: lbvh_gpu.hip — and staged in LDS by the traversal kernels (rt_device.h)
constexpr int kStackSize = 64;      // traversal stack entries per ray; builders keep depth < 64
constexpr uint32_t kNoTri = 0xFFFFFFFFu;
constexpr float kTNear = 0.0001f;   // RayData::to_embree tnear (src/camera.hpp:50)

// ---- triangles, leaf order: 48 bytes = 3 x float4 -------------------------------------------------
//   a = (v0.x, v0.y, v0.z, e1.x)  b = (e1.y, e1.z, e2.x, e2.y)  c = (e2.z, bits(global index), 0, 0)
struct alignas(16) TriRec {
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t global_index; // triangle index in rt_scene_desc order (tie-break + shading lookup)
    uint32_t pad[2];
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");
// What the device reads per triangle: the record's ten live dwords, packed (the two padding words were 17 % of the
// triangle traffic): bytes 0-15 (v0.xyz, e1.x), 16-31 (e1.yz, e2.xy), 32-39 (e2.z, global index) at a 40-byte stride (8-byte aligned).
constexpr uint32_t kTriBytes = 40;

// ---- per-triangle shading attributes, GLOBAL order: 64 bytes ------------------------------------
// What trace_ray gathers through index_buffer/normal_buffer/uv_buffer (src/trace_ray.hpp:34-45),
// pre-gathered per triangle so a hit costs one aligned 64-byte fetch instead of a dependent chain.
// `instance`: the instance index — and, in scenes with at most 2^20 instances and 2^12 materials (SceneDev::packed_mat), the instance's
// material index in bits 20..31 as well: shading then fetches the material record side by side with the instance record instead of
// after it (one hop less in the chain shade record -> instance -> material -> texel).
constexpr uint32_t kPackedInstBits = 20, kPackedInstMask = (1u << kPackedInstBits) - 1u;
struct alignas(64) ShadeRec {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    uint32_t instance;
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec must be 64 bytes");

// GeometryData::obj_to_world + material id (src/scene.hpp:17-24): 48 bytes.
// Where the shading word is packed (above) the device table holds every DISTINCT normal matrix once, the one most triangles use first
// (scene_build.cpp): a glTF scene has one instance per node and primitive and a handful of different matrices — the atrium 166 and 5 —
// and shading reads the first kLdsNm of them, with the first kLdsMats materials, from LDS (rt_device.h: ShadeTables).
// (A CU's 160 KB of LDS are handed out in 1,280-byte granules — measured: k_megakernel's three workgroups per CU fit up to 53,760 bytes each,
// profiles/r05_lds_granule.txt — which leaves 1,216 bytes beside the staged tree top, the stacks and the colour sums: 8 x 48 + 24 x 32 = 1,152.)
constexpr uint32_t kLdsNm = 8, kLdsMats = 24;
struct alignas(16) InstRec {
    float normal_mat[9]; // column-major 3x3
    uint32_t material;
    uint32_t pad[2];
};
static_assert(sizeof(InstRec) == 48, "InstRec must be 48 bytes");

// Material (src/material.hpp:163-238): 48 bytes
struct alignas(16) MatRec {
    uint32_t type;
    uint32_t tex_kind;
    float color[3];
    uint32_t tex_layer;
    float emissive[3];
    float roughness;
    float ior;
    uint32_t pad;
};
static_assert(sizeof(MatRec) == 48, "MatRec must be 48 bytes");

// device pointers of one scene replica
struct SceneDev {
    const BvhNode* nodes;
    const uint8_t* tris; // leaf order, kTriBytes per triangle
    const ShadeRec* shade;
    const InstRec* inst;
    const MatRec* mats;
    const uint8_t* tex; // n_layers x 512 x 512 x 4
    uint32_t n_nodes;
    uint32_t n_tris;
    float sky[3];
    float cell_lo[3], cell_scale[3]; // ray re-ordering (f-3): cell = (origin - cell_lo) * cell_scale, 0..4 per axis over the scene's bounds
    uint32_t packed_mat;             // 1: ShadeRec::instance carries the material index in bits 20..31
    uint32_t lds_nm, lds_mats;       // normal matrices / materials the shading kernels stage in LDS (0: shading word not packed)
};

// == Camera POD (src/camera.hpp:65-72)
struct CameraDev {
    float center[3], pixel00[3], du[3], dv[3];
    int32_t width, height;
};

// image-space tile owned by this renderer (strips of `strip_rows` rows, strip k -> rank k % world)
struct TileDev {
    uint32_t rank, world, strip_rows;
    int32_t local_rows;
};

} // namespace rt
