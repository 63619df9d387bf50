// rt_device.h — gfx950 device functions of the ray-trace hot path.
//
// Each function names the reference code it stands for (file:line under the reference tree).
// Arithmetic contract: single IEEE-754 binary32 operations evaluated as written,
// left to right, no FMA contraction (this TU is compiled with -ffp-contract=off), correctly
// rounded division and sqrt (-fhip-fp32-correctly-rounded-divide-sqrt), half storage with
// round-to-nearest-even. The ONLY place that uses fused/approximate arithmetic is the BVH box
// test, which merely culls (boxes are padded) and cannot change a result.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rt_mi355x.h"
#include "rt_types.h"

namespace rt {

#define RT_DEV __device__ __forceinline__

struct f3 {
    float x, y, z;
};
RT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// ---- correctly rounded 1/x and 1/sqrt(x) in a few fma (round 3) ------------------------------------------------------------
// The contract wants RN(1 / x) and RN(1 / RN(sqrt(x))) bit for bit. hipcc's IEEE expansions of those are 11 and 28
// dependent instructions (v_div_scale x 2, v_rcp, five fma, v_div_fmas, v_div_fixup; the square root with its denormal scaling, two
// neighbour tests and four selects): a third of a shading round's instructions, in one serial chain. On gfx950
//     y = v_rcp_f32(x);  y + y * (1 - x * y)                                          (two fma)
// IS RN(1 / x) for EVERY x whose biased exponent lies in 1..252, either sign, and
//     r = v_rsq_f32(x);  g = x r, h = r / 2;  e = 1/2 - h g;  g += g e, h += h e;  g + h * (x - g g)        (Goldschmidt + one residual)
// IS RN(sqrt(x)) for every x with biased exponent 25..254 — checked on ALL 2^32 inputs (a search over candidate forms in round 3 chose these;
// rt_probe_rounding runs the check on the functions below, in the GPU test suite: tests/test_gpu_parity.py). Outside those ranges
// (zero, denormals, huge, inf, NaN, negative radicands) a wave-uniform branch takes the compiler's IEEE sequence, so the functions
// equal the IEEE expressions everywhere; inside, a normalisation costs 14 instructions instead of 28 + 11.
RT_DEV float rcp_rn(float x) { // == 1.0f / x
    const uint32_t t = __float_as_uint(x) + __float_as_uint(x); // the sign shifted out
    const bool ok = t - 0x01000000u < 0xFC000000u;              // biased exponent 1 .. 252
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) == 0ull, 1)) {
        const float y = __builtin_amdgcn_rcpf(x);
        return __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
    }
    return 1.0f / x;
}
RT_DEV float inv_sqrt2(float x) { // == 1.0f / sqrtf(x): two roundings, as glm::normalize / sycl::normalize
    const bool ok = __float_as_uint(x) - 0x0C800000u < 0x73000000u; // positive, biased exponent 25 .. 254
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) == 0ull, 1)) {
        const float r = __builtin_amdgcn_rsqf(x);
        float g = x * r, h = 0.5f * r;
        const float e = __builtin_fmaf(-h, g, 0.5f);
        g = __builtin_fmaf(g, e, g), h = __builtin_fmaf(h, e, h);
        const float s = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g); // RN(sqrt(x)), exponent 76 .. 191: inside rcp_rn's range
        const float y = __builtin_amdgcn_rcpf(s);
        return __builtin_fmaf(__builtin_fmaf(-s, y, 1.0f), y, y);
    }
    return 1.0f / __builtin_sqrtf(x);
}
// glm::normalize / sycl::normalize: v * (1 / sqrt(dot(v,v)))
RT_DEV f3 normalize3(f3 a) {
    const float inv = inv_sqrt2(dot3(a, a));
    return mk3(a.x * inv, a.y * inv, a.z * inv);
}
RT_DEV float clamp01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }

// sycl::half storage of dir / attenuation / radiance (src/camera.hpp:18-43)
RT_DEV uint16_t f2h(float f) {
    // The value must be ROUNDED TO f32 FIRST and then to f16 (two roundings, as the reference's
    // float -> sycl::half assignment does). Without this barrier hipcc folds "(half)(a*b)" into
    // v_fma_mixlo_f16, which rounds the exact product once and differs in ~1e-4 of the cases.
    asm("" : "+v"(f));
    _Float16 h = (_Float16)f; // v_cvt_f16_f32, round-to-nearest-even
    return __builtin_bit_cast(uint16_t, h);
}
RT_DEV float h2f(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }

// ---- XorShift32State (src/xorshift.hpp:8-49) -------------------------------------------------------
RT_DEV float rng_next(uint32_t& a) { // operator()(): :11-20
    uint32_t x = a;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    a = x;
    return (float)x * (1.0f / 4294967296.0f);
}
RT_DEV float rng_range(uint32_t& a, float mn, float mx) { return mn + (mx - mn) * rng_next(a); } // :22-24
RT_DEV f3 rng_unit_vector(uint32_t& a) { // vec(-1,1) then normalize: :30-40, draws in x,y,z order
    float x = rng_range(a, -1.0f, 1.0f);
    float y = rng_range(a, -1.0f, 1.0f);
    float z = rng_range(a, -1.0f, 1.0f);
    return normalize3(mk3(x, y, z));
}

// per-pixel seed: std::hash<size_t> is the identity (src/render_wavefront.cpp:69-73,
// src/render_megakernel.cpp:90-93,144-146)
RT_DEV uint32_t pixel_seed(int x, int y, int w, int h, uint32_t seed_mode) {
    if (seed_mode == RT_SEED_MEGAKERNEL) {
        uint32_t h8 = (uint32_t)((h + 7) / 8) * 8u;
        return (uint32_t)x * h8 + (uint32_t)y;
    }
    return (uint32_t)x + (uint32_t)y * (uint32_t)w;
}

// local (tile) row -> global image row
RT_DEV int tile_global_row(const TileDev& t, int local_row) {
    uint32_t strip = (uint32_t)local_row / t.strip_rows;
    uint32_t within = (uint32_t)local_row % t.strip_rows;
    return (int)((strip * t.world + t.rank) * t.strip_rows + within);
}

// ---- ray state: RayData (src/camera.hpp:12-44) -----------------------------------------------------
struct RayState {
    f3 org;
    uint16_t dir[3], att[3], rad[3];
};

// Camera::get_ray + pixel_sample_square (src/camera.hpp:109-131) and the RayData ctor (:30-44)
RT_DEV RayState camera_ray(const CameraDev& c, int x, int y, uint32_t& rng) {
    f3 p00 = mk3(c.pixel00[0], c.pixel00[1], c.pixel00[2]);
    f3 du = mk3(c.du[0], c.du[1], c.du[2]);
    f3 dv = mk3(c.dv[0], c.dv[1], c.dv[2]);
    f3 center = mk3(c.center[0], c.center[1], c.center[2]);
    f3 pixel_center = (p00 + ((float)x * du)) + ((float)y * dv);
    float px = -0.5f + rng_next(rng);
    float py = -0.5f + rng_next(rng);
    f3 jitter = (px * du) + (py * dv);
    f3 pixel_sample = pixel_center + jitter;
    f3 d = pixel_sample - center;
    RayState r;
    r.org = center;
    r.dir[0] = f2h(d.x), r.dir[1] = f2h(d.y), r.dir[2] = f2h(d.z);
    r.att[0] = r.att[1] = r.att[2] = 0x3C00; // half(1.0)
    r.rad[0] = r.rad[1] = r.rad[2] = 0;      // half(0.0)
    return r;
}

// ---- closest hit: replaces rtcIntersect1 (src/trace_ray.hpp:18-27) ---------------------------------
struct Hit {
    float t, u, v;
    uint32_t tri; // global triangle index, kNoTri on a miss
};

// Moller-Trumbore on the precomputed (v0, e1, e2); barycentrics in Embree's convention
// P = (1-u-v) v0 + u v1 + v v2. Edge tests on the sign-corrected numerators; one division per
// accepted candidate. Minimum t wins, equal t -> lowest global index (order independent).
// (Round 4) The test's cross and dot products are FUSED: every a * b + c of them is one v_fma_f32 with one rounding (dot3f / cross3f; the
// CPU oracle states the same expressions). This is the one place where that is legitimate: the test stands for Embree's, whose
// bits nobody can pin, so its arithmetic is this repository's to define — 36 instead of 50 arithmetic instructions per test.
RT_DEV float dot3f(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RT_DEV f3 cross3f(f3 a, f3 b) {
    return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
RT_DEV void tri_test_regs(float4 a, float4 b, float2 c, f3 o, f3 d, Hit& best) { // the record's 10 live dwords
    const f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
    const uint32_t gidx = __float_as_uint(c.y);
    const f3 p = cross3f(d, e2);
    const float det = dot3f(e1, p);
    if (!(det != 0.0f)) return;
    const f3 tv = o - v0;
    const f3 q = cross3f(tv, e1);
    const float sgn = det < 0.0f ? -1.0f : 1.0f;
    const float un0 = dot3f(tv, p), vn0 = dot3f(d, q);
    const float un = un0 * sgn, vn = vn0 * sgn;
    if (!(un >= 0.0f && vn >= 0.0f && un + vn <= __builtin_fabsf(det))) return;
    const float inv = rcp_rn(det);
    const float t = dot3f(e2, q) * inv;
    if (!(t > kTNear)) return;
    if (t < best.t || (t == best.t && gidx < best.tri)) {
        best.t = t;
        best.u = un0 * inv;
        best.v = vn0 * inv;
        best.tri = gidx;
    }
}
// the packed 40-byte device records are only 8-byte aligned: the loads say so (still one dwordx4 each: gfx950 needs dword alignment only)
typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef float f2a8 __attribute__((ext_vector_type(2), aligned(8)));
RT_DEV float4 tri_ld4(const uint8_t* p) {
    const f4a8 v = *reinterpret_cast<const f4a8*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
RT_DEV float2 tri_ld2(const uint8_t* p) {
    const f2a8 v = *reinterpret_cast<const f2a8*>(p);
    return make_float2(v.x, v.y);
}
RT_DEV void tri_test(const uint8_t* __restrict__ tr, f3 o, f3 d, Hit& best) {
    tri_test_regs(tri_ld4(tr), tri_ld4(tr + 16), tri_ld2(tr + 32), o, d, best);
}

// ---- BVH4 traversal as a resumable per-lane state machine --------------------------------------------
// One ray per lane. A lane is in one of three states, encoded in `cur`:
//   cur >= 0            at an inner node (index `cur`)
//   cur <  0, != DONE   at a leaf: ~cur = (first << 2) | (remaining - 1), triangles tested ONE per step
//   cur == kTravDone    traversal finished, `best` holds the closest hit
// The WAVE executes one uniform step type per iteration (trav_step_wave): either every lane that is at
// an inner node takes an inner step, or every lane that is at a leaf tests one triangle — whichever
// class currently holds more lanes. Minority lanes wait a few iterations instead of dragging the whole
// wave through a divergent branch; measured VALU lane utilisation of the first version (if/else inside
// one loop) was 27 %. The box test is a padded-box cull: it uses fma and an approximate reciprocal on
// purpose (never visible in results).
constexpr int32_t kTravDone = (int32_t)0x80000000;
struct Trav {
    f3 o, d;              // ray (d unnormalised, as stored)
    float ix, iy, iz;     // 1/d (|d| clamped away from 0)
    float ox, oy, oz;     // -o/d
    Hit best;
    int32_t cur;
    uint32_t sp; // LDS byte address of the lane's next free stack entry (TravStack::base = empty); beyond the LDS part it keeps counting
};

struct TravStack;
RT_DEV uint32_t stack_base(const TravStack& st);
RT_DEV void trav_begin(Trav& T, f3 o, f3 d, const TravStack& st) {
    T.o = o, T.d = d;
    T.best.t = __builtin_huge_valf();
    T.best.u = T.best.v = 0.0f;
    T.best.tri = kNoTri;
    // A zero direction component would give inv = inf and, in the fma form b*inv - o*inv, inf - inf = NaN
    // on the far side of the slab; clamp |d| to 1e-30 so the product stays finite (|b|, |o| < 1e8).
    const float kTiny = 1e-30f;
    const float dx = __builtin_fabsf(d.x) < kTiny ? __builtin_copysignf(kTiny, d.x) : d.x;
    const float dy = __builtin_fabsf(d.y) < kTiny ? __builtin_copysignf(kTiny, d.y) : d.y;
    const float dz = __builtin_fabsf(d.z) < kTiny ? __builtin_copysignf(kTiny, d.z) : d.z;
    T.ix = __builtin_amdgcn_rcpf(dx), T.iy = __builtin_amdgcn_rcpf(dy), T.iz = __builtin_amdgcn_rcpf(dz);
    T.ox = -o.x * T.ix, T.oy = -o.y * T.iy, T.oz = -o.z * T.iz;
    T.cur = 0; // root
    T.sp = stack_base(st); // empty stack
}

// Traversal stack: the first kLdsStack entries of every lane live in LDS (entry-major, lane-minor:
// conflict-free ds_read/ds_write_b32), deeper entries spill to per-lane scratch (rare). The first
// version kept the whole stack in scratch: every push/pop was an L1 access and a write to HBM.
// The stack pointer IS the LDS byte address of the next free entry (round 2): a push is ds_write + one v_add_u32, a pop one
// v_sub_u32 + ds_read, with no index -> address arithmetic (v_lshl_or_b32 is a 4-cycle instruction) and no per-push depth test:
// a step asks ONCE, wave-uniformly, whether every lane's next three pushes and its pop stay inside the LDS part (they nearly
// always do) and only otherwise takes the general routines, which turn the address back into an entry index.
// (Round 4) Nothing on the common path branches: entry 0 of every lane's stack is a sentinel that holds kTravDone, so a pop needs no
// test for an empty stack, and an inner step WRITES its three candidate words to the next free entries in any case — only the stack
// pointer depends on the hits (trav_inner). Each removed exec-mask save / branch / restore was worth ~1 % of the frame.
constexpr int kLdsStack = 12;
typedef __attribute__((address_space(3))) int32_t lds_i32; // forces ds_read/ds_write (never flat_*)
struct TravStack {
    uint32_t base;  // LDS byte address of this lane's entry 0 (the sentinel)
    int32_t* spill; // kStackSize - kLdsStack entries of scratch
    uint32_t pitch; // bytes from one entry of a lane to its next: 4 * threads per block (a power of two)
};
RT_DEV uint32_t stack_base(const TravStack& st) { return st.base + st.pitch; } // the first entry above the sentinel
RT_DEV lds_i32* lds_at(uint32_t addr) { return (lds_i32*)(size_t)addr; }
// true when the lane's next `pushes` pushes and a pop all stay inside the LDS part of its stack
RT_DEV bool stack_shallow(const TravStack& st, const Trav& T, uint32_t pushes) { return T.sp <= st.base + ((uint32_t)kLdsStack - pushes) * st.pitch; }
RT_DEV void trav_pop_lds(Trav& T, const TravStack& st) { // caller: stack_shallow
    T.sp -= st.pitch; // entry 0 holds kTravDone (RT_TRAVERSAL_LDS): popping an empty stack ends the traversal with no test of its own
    T.cur = *lds_at(T.sp);
}
RT_DEV void stk_push(const TravStack& st, Trav& T, int32_t v) { // any depth
    const uint32_t e = (T.sp - st.base) / st.pitch;
    if (e < (uint32_t)kLdsStack) *lds_at(T.sp) = v;
    else st.spill[e - (uint32_t)kLdsStack] = v;
    T.sp += st.pitch;
}
RT_DEV void trav_pop(Trav& T, const TravStack& st) { // any depth
    T.sp -= st.pitch;
    const uint32_t e = (T.sp - st.base) / st.pitch;
    T.cur = e < (uint32_t)kLdsStack ? *lds_at(T.sp) : st.spill[e - (uint32_t)kLdsStack];
}

// The first kTopNodes nodes of the node array (the top of the BVH4: scene_build.cpp orders them by surface area, ~5 levels) are staged in
// LDS by every workgroup of the traversal kernels, as four structure-of-arrays planes of 16 bytes per node (the node's
// four 16-byte words), so that the 16-byte reads of different nodes spread over the banks. Measured on the atrium: 9.5 of the 17.9 inner-node
// visits of an average ray fall into these nodes; taking them out of the vector L1 path (the first
// versions were bound by L1 tag throughput: ~1 lane-load per cycle and CU) is worth more than any ALU saving.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
struct TopTree {
    lds_u32x4* w0; // origin.xyz, scale.x
    lds_u32x4* w1; // qlo_x, qhi_x, qlo_y, qhi_y
    lds_u32x4* w2; // qlo_z, qhi_z, scale.y, scale.z
    lds_u32x4* ch; // child[4]
    int32_t count; // nodes staged (0 = none)
};
// cooperative fill by the whole workgroup; the caller synchronises afterwards
RT_DEV void top_tree_fill(const SceneDev& S, u32x4* w0, u32x4* w1, u32x4* w2, u32x4* ch, int32_t count) {
    for (int32_t i = (int32_t)threadIdx.x; i < count; i += (int32_t)blockDim.x) {
        const u32x4* np = reinterpret_cast<const u32x4*>(S.nodes + i);
        w0[i] = np[0];
        w1[i] = np[1];
        w2[i] = np[2];
        ch[i] = np[3];
    }
}

// ---- instruction selection helpers (profiles/r02_valu_calibration.json) ------------------------------------------------
// The traversal kernels are bound by VALU issue cycles, and on gfx950 a select is cheap or dear by its ENCODING: v_cndmask_b32_e64
// on an SGPR pair costs 4 cycles, the VOP2 form on VCC 16 unless it directly follows the compare that wrote VCC. hipcc picks either;
// these helpers pin the compare to an SGPR-pair result (ballot) and the select to the e64 form.
typedef unsigned long long lmask;
RT_DEV lmask lanes(bool c) { return __builtin_amdgcn_ballot_w64(c); } // v_cmp_*_e64 into an SGPR pair
RT_DEV float sel(lmask m, float a, float b) { // m ? b : a
    float d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m));
    return d;
}
RT_DEV uint32_t sel(lmask m, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m));
    return d;
}
RT_DEV int32_t sel(lmask m, int32_t a, int32_t b) { return (int32_t)sel(m, (uint32_t)a, (uint32_t)b); }

// The direction signs of the wave's 64 rays as three lane masks. A ray's signs do not change while it is traversed, so the kernels
// take them ONCE per traversal phase, in wave-uniform control flow after the shading round that started the rays (trav_signs), and
// every inner step selects its near / far plane words on these SGPR pairs directly. (Round 3: written as lanes(T.ix < 0.0f) inside the
// step, the compiler hoisted the compares out of the loop by itself and then rebuilt each mask per step with v_cndmask 0/1 + v_cmp_ne —
// six half-rate instructions, 24 of the step's ~530 issue cycles, for three masks it already held.) Bits of lanes without a ray are
// arbitrary and never used.
struct TravSigns {
    lmask nx, ny, nz;
};

// byte k of a word as float (v_cvt_f32_ubyteK)
RT_DEV float ub0(uint32_t w) { return (float)(w & 0xffu); }
RT_DEV float ub1(uint32_t w) { return (float)((w >> 8) & 0xffu); }
RT_DEV float ub2(uint32_t w) { return (float)((w >> 16) & 0xffu); }
RT_DEV float ub3(uint32_t w) { return (float)(w >> 24); }

// wave-level scheduling statistics (diagnostic builds of the kernels only: RT_KERNEL_STATS=1)
struct WaveStats {
    uint32_t inner_steps = 0, inner_lanes = 0, leaf_steps = 0, leaf_lanes = 0;
    uint32_t shade_rounds = 0, shade_lanes = 0, live_lane_steps = 0, top_lanes = 0;
};
// inner step: fetch the 64-byte BVH4 node, decode and test its four quantised child boxes, descend into
// the nearest hit child and push the other hits far-to-near.
//   t(plane q) = (origin + q*s - o) * inv = q * (s*inv) + (origin*inv - o*inv)
// The near/far plane words are picked once per axis from the ray's direction sign, so no per-child
// min/max pairing is needed.
RT_DEV TravSigns trav_signs(const Trav& T) { return TravSigns{lanes(T.ix < 0.0f), lanes(T.iy < 0.0f), lanes(T.iz < 0.0f)}; }
RT_DEV void trav_inner(const SceneDev& S, Trav& T, const TravStack& stack, const TopTree& top, const TravSigns& sg) {
    const float inf = __builtin_huge_valf();
    float k0, k1, k2, k3;
    u32x4 w0, w1, w2, chw;
    if (T.cur < top.count * 64) { // top of the tree: LDS, 16 bytes per node and plane
        const uint32_t o = (uint32_t)T.cur >> 2;
        w0 = *(lds_u32x4*)(size_t)((uint32_t)(size_t)top.w0 + o), w1 = *(lds_u32x4*)(size_t)((uint32_t)(size_t)top.w1 + o);
        w2 = *(lds_u32x4*)(size_t)((uint32_t)(size_t)top.w2 + o), chw = *(lds_u32x4*)(size_t)((uint32_t)(size_t)top.ch + o);
    } else {
        // The four 16-byte words of a node in global memory, issued from inline assembly so that hipcc's wait-count pass does not see them:
        // it would otherwise hold the LDS reads of the other lanes back until these loads have RETURNED (s_waitcnt vmcnt(0) in front of the
        // ds_read_b128s — both write the same registers, and the pass does not reason about disjoint exec masks), which adds the LDS
        // latency to the global one in every step that has lanes of both kinds. The hardware writes only the lanes each instruction was
        // issued for. The waits these loads need are the two explicit ones below the join (loads return in order: three words, then the fourth).
        asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                     "global_load_dwordx4 %1, %4, %5 offset:16\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:32\n\t"
                     "global_load_dwordx4 %3, %4, %5 offset:48"
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(chw)
                     : "v"((uint32_t)T.cur), "s"(S.nodes)
                     : "memory");
    }
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(w0), "+v"(w1), "+v"(w2)); // the child words (the fourth load) are not needed before the sort
    const float ax = __uint_as_float(w0.w) * T.ix;
    const float ay = __uint_as_float(w2.z) * T.iy;
    const float az = __uint_as_float(w2.w) * T.iz;
    const float bx = __builtin_fmaf(__uint_as_float(w0.x), T.ix, T.ox);
    const float by = __builtin_fmaf(__uint_as_float(w0.y), T.iy, T.oy);
    const float bz = __builtin_fmaf(__uint_as_float(w0.z), T.iz, T.oz);
    const lmask nx = sg.nx, ny = sg.ny, nz = sg.nz;
    const uint32_t qnx = sel(nx, w1.x, w1.y), qfx = sel(nx, w1.y, w1.x);
    const uint32_t qny = sel(ny, w1.z, w1.w), qfy = sel(ny, w1.w, w1.z);
    const uint32_t qnz = sel(nz, w2.x, w2.y), qfz = sel(nz, w2.y, w2.x);
    // An absent child needs no test of its own: its box is stored inverted (qlo = 255, qhi = 0), which
    // makes entry > exit on every axis for either direction sign. (Pairing children into v_pk_fma_f32 costs
    // 4 cycles per pair, the same as two v_fma_f32.)
    // The inversion holds for every ray inside the contract's range (origins within 100 scene diameters, include/rt_mi355x.h:
    // rt_intersect_batch). It would take an origin more than ~660 diameters away for 255 * scale * inv to be absorbed by the rounding
    // of origin * inv - o * inv on all three axes at once (entry == exit), and out there the padded boxes are no longer conservative
    // for any child, absent or not — so the child words are not tested here (an explicit kChildEmpty test per slot was measured at
    // 12 issue cycles of ~500 per step).
#define RT_CHILD(K, CVT)                                                                                  \
    {                                                                                                    \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(CVT(qnx), ax, bx), __builtin_fmaf(CVT(qny), ay, by)), \
                                         __builtin_fmaxf(__builtin_fmaf(CVT(qnz), az, bz), 0.0f));       \
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf(CVT(qfx), ax, bx), __builtin_fmaf(CVT(qfy), ay, by)), \
                                         __builtin_fminf(__builtin_fmaf(CVT(qfz), az, bz), T.best.t));   \
        K = sel(lanes(tn <= tf), inf, tn);                                                               \
    }
    RT_CHILD(k0, ub0)
    RT_CHILD(k1, ub1)
    RT_CHILD(k2, ub2)
    RT_CHILD(k3, ub3)
#undef RT_CHILD
    // sorting network on (entry distance, child); misses carry +inf and sink to the end. One comparator = one compare into an
    // SGPR pair + four e64 selects on it (18 cycles).
#define RT_CE(KA, CA, KB, CB)                                                        \
    {                                                                               \
        const lmask sw = lanes(KB < KA);                                            \
        const float ka = sel(sw, KA, KB), kb = sel(sw, KB, KA);                     \
        const int32_t ca = sel(sw, CA, CB), cb = sel(sw, CB, CA);                   \
        KA = ka, KB = kb, CA = ca, CB = cb;                                         \
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(chw));
    int32_t c0 = (int32_t)chw.x, c1 = (int32_t)chw.y, c2 = (int32_t)chw.z, c3 = (int32_t)chw.w;
    RT_CE(k0, c0, k1, c1)
    RT_CE(k2, c2, k3, c3)
    RT_CE(k0, c0, k2, c2)
    // with fewer than five comparators the hits are no longer a sorted prefix of (k1, k2, k3): each is tested on its own below
#undef RT_CE
    const bool descend = k0 < inf;
    if (lanes(!stack_shallow(stack, T, 3u)) == 0ull) { // every lane of this step: LDS only (one wave-uniform test, no per-push ones)
        // pushes without branches: the word is written to the next free entry in any case (there is room for three, asked above), and
        // only the stack pointer depends on the hit (three conditional pushes: three exec-mask regions with a branch each, -0.9 %)
        *lds_at(T.sp) = c3; // farthest first (exactly so with the full network)
        T.sp = sel(lanes(k3 < inf), T.sp, T.sp + stack.pitch);
        *lds_at(T.sp) = c2;
        T.sp = sel(lanes(k2 < inf), T.sp, T.sp + stack.pitch);
        *lds_at(T.sp) = c1;
        T.sp = sel(lanes(k1 < inf), T.sp, T.sp + stack.pitch);
        if (descend) T.cur = c0;
        else trav_pop_lds(T, stack);
    } else {
        if (k3 < inf) stk_push(stack, T, c3);
        if (k2 < inf) stk_push(stack, T, c2);
        if (k1 < inf) stk_push(stack, T, c1);
        if (descend) T.cur = c0;
        else trav_pop(T, stack);
    }
}

// leaf step. BATCH = false: ONE triangle of the leaf per step, then the next one or pop (rt_intersect_batch).
// BATCH = true (both renderers): the whole leaf (1-2 triangles) in this step — a lane at a leaf waits for the leaf class to win
// the wave's vote, so every extra leaf step costs it several iterations. Measured on the megakernel: 12.5 M -> 6.2 M leaf steps
// per 4-spp frame, +2.5 % rays/s; EXTEND: +6 % (round 2). Round 4: no loop and no masked loads any more (below): +4.8 %.
template <bool BATCH>
RT_DEV void trav_leaf(const SceneDev& S, Trav& T, const TravStack& stack) {
    const uint32_t code = (uint32_t)~T.cur;
    const uint32_t first = code >> 2, rem = code & 3u;
    if (BATCH) {
        static_assert(kMaxLeafTris == 2 && kTriBytes == 40, "the whole-leaf step reads a leaf as 80 contiguous bytes");
        // The two records of a leaf are 80 contiguous bytes: FIVE 16-byte loads instead of 2 x (16 + 16 + 8) — one look-up less in the vector
        // L1 per leaf, the path this kernel loads most — and all five UNCONDITIONAL: a leaf of one record (2 % of the SAH
        // leaves, every LBVH leaf) reads 40 bytes of its neighbour and ignores them (the buffer ends in 48 bytes of padding, rt_abi.hip:
        // rt_scene_create). With the second record's loads under `if (two)` the first test waited for all five at the join and eight
        // v_movs initialised the merged registers: 134.1 instead of 126.5 ms per frame (EXPERIMENTS.md).
        const uint8_t* p4 = S.tris + (size_t)first * kTriBytes;
        const float4 l0 = tri_ld4(p4), l1 = tri_ld4(p4 + 16), l2 = tri_ld4(p4 + 32), l3 = tri_ld4(p4 + 48), l4 = tri_ld4(p4 + 64);
        const bool two = rem != 0u;
        tri_test_regs(l0, l1, make_float2(l2.x, l2.y), T.o, T.d, T.best);
        if (__ballot(two) != 0ull) {
            if (two) tri_test_regs(make_float4(l2.z, l2.w, l3.x, l3.y), make_float4(l3.z, l3.w, l4.x, l4.y), make_float2(l4.z, l4.w), T.o, T.d, T.best);
        }
        if (lanes(!stack_shallow(stack, T, 0u)) == 0ull) trav_pop_lds(T, stack);
        else trav_pop(T, stack);
    } else {
        tri_test(S.tris + (size_t)first * kTriBytes, T.o, T.d, T.best);
        if (rem) T.cur = ~(int32_t)(((first + 1u) << 2) | (rem - 1u));
        else trav_pop(T, stack);
    }
}

// One wave-uniform step. Returns the number of lanes still traversing BEFORE the step (0 = all done).
template <bool STATS = false, bool LEAF_BATCH = false>
RT_DEV uint32_t trav_step_wave(const SceneDev& S, Trav& T, const TravStack& stack, const TopTree& top, const TravSigns& sg, WaveStats* ws = nullptr) {
    const bool inner = T.cur >= 0;
    const bool leaf = T.cur < 0 && T.cur != kTravDone;
    const uint32_t ni = (uint32_t)__popcll(__ballot(inner)), nl = (uint32_t)__popcll(__ballot(leaf));
    // the vote, weighted 3 : 4 — a leaf step as soon as the leaf lanes exceed 3/4 of the inner lanes: a lane parked at a leaf is idle for as
    // many iterations as the vote goes against it (plain majority: -1.2 %; EXPERIMENTS.md)
    if (ni * 3u >= nl * 4u && !(LEAF_BATCH && nl >= 64u)) {
        if (STATS && ni) ws->inner_steps++, ws->inner_lanes += ni, ws->top_lanes += (uint32_t)__popcll(__ballot(inner && T.cur < kTopNodes * 64));
        if (inner) trav_inner(S, T, stack, top, sg);
    } else {
        if (STATS) ws->leaf_steps++, ws->leaf_lanes += (uint32_t)__popcll(__ballot(leaf));
        if (leaf) trav_leaf<LEAF_BATCH>(S, T, stack);
    }
    return ni + nl;
}

// run to completion (rt_intersect_batch; lanes that finish early wait for the wave)
RT_DEV Hit intersect(const SceneDev& S, f3 o, f3 d, const TravStack& stack, const TopTree& top) {
    Trav T;
    trav_begin(T, o, d, stack);
    const TravSigns sg = trav_signs(T);
    while (trav_step_wave(S, T, stack, top, sg)) {}
    return T.best;
}

// ---- materials (src/material.hpp, src/util.hpp:82-125) ----------------------------------------------
RT_DEV bool near_zero3(f3 e) { // src/util.hpp:103-107
    const float s = 1e-8f;
    return (__builtin_fabsf(e.x) < s) && (__builtin_fabsf(e.y) < s) && (__builtin_fabsf(e.z) < s);
}
RT_DEV float length_squared3(f3 v) { // src/util.hpp:109-112: length(v) * length(v)
    float l = __builtin_sqrtf(dot3(v, v));
    return l * l;
}
RT_DEV f3 reflect3(f3 v, f3 n) { return v - (2.0f * dot3(v, n)) * n; } // src/util.hpp:114-116
RT_DEV f3 refract3(f3 uv, f3 n, float etai_over_etat) {                // src/util.hpp:118-125
    float cos_theta = __builtin_fminf(dot3(-uv, n), 1.0f);
    f3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    f3 r_out_parallel = (-__builtin_sqrtf(__builtin_fabsf(1.0f - length_squared3(r_out_perp)))) * n;
    return r_out_perp + r_out_parallel;
}
// MaterialDielectric::reflectance (src/material.hpp:120-125); pow(x,5) = ((x*x)*(x*x))*x
RT_DEV float schlick(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x5 = (x2 * x2) * x;
    return r0 + (1.0f - r0) * x5;
}
// b / 255.0f for b = 0..255 without the ten-instruction IEEE division: one Newton step on b * RN(1/255) with exact
// remainders (two fmas) is correctly rounded for all 256 inputs (enumerated in the CPU tests, and checked bit for bit
// through a texture that holds every value in tests/test_gpu_parity.py)
RT_DEV float unorm8_to_float(uint32_t b) {
    const float x = (float)b, r = 1.0f / 255.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

// Texture::sample (src/material.hpp:45-53): constant colour, or nearest / repeat / normalised
// fetch from layer tex_layer of the 512x512 RGBA8 array (sampler: src/render_wavefront.cpp:193-197)
// ... in two halves, so that shading can issue the texel fetch (a miss into the Infinity Cache more often than not: 9 MB of texels,
// nearest, used once) as soon as the hit's uv is known and interpolate / normalise / draw while it is in flight, instead of reaching
// it at the end of Material::scatter's branch with nothing left to do but wait (round 3)
RT_DEV bool albedo_is_texture(const MatRec& m) { return (m.type == RT_MAT_DIFFUSE || m.type == RT_MAT_METALLIC) && m.tex_kind != RT_TEX_COLOR; }
RT_DEV uint32_t albedo_fetch(const SceneDev& S, const MatRec& m, float u, float v) { // caller: albedo_is_texture(m)
    float fu = u - __builtin_floorf(u);
    float fv = v - __builtin_floorf(v);
    int iu = (int)__builtin_floorf(fu * 512.0f);
    int iv = (int)__builtin_floorf(fv * 512.0f);
    iu = iu < 0 ? 0 : (iu > 511 ? 511 : iu);
    iv = iv < 0 ? 0 : (iv > 511 ? 511 : iv);
    return *reinterpret_cast<const uint32_t*>(S.tex + (((size_t)m.tex_layer * 512u + (size_t)iv) * 512u + (size_t)iu) * 4u);
}
RT_DEV f3 albedo_decode(uint32_t texel) {
    return mk3(unorm8_to_float(texel & 0xffu), unorm8_to_float((texel >> 8) & 0xffu), unorm8_to_float((texel >> 16) & 0xffu));
}
RT_DEV f3 albedo(const SceneDev& S, const MatRec& m, float u, float v) {
    if (m.tex_kind == RT_TEX_COLOR) return mk3(m.color[0], m.color[1], m.color[2]);
    float fu = u - __builtin_floorf(u);
    float fv = v - __builtin_floorf(v);
    int iu = (int)__builtin_floorf(fu * 512.0f);
    int iv = (int)__builtin_floorf(fv * 512.0f);
    iu = iu < 0 ? 0 : (iu > 511 ? 511 : iu);
    iv = iv < 0 ? 0 : (iv > 511 ? 511 : iv);
    const uint32_t texel = *reinterpret_cast<const uint32_t*>(
        S.tex + (((size_t)m.tex_layer * 512u + (size_t)iv) * 512u + (size_t)iu) * 4u);
    return mk3(unorm8_to_float(texel & 0xffu), unorm8_to_float((texel >> 8) & 0xffu), unorm8_to_float((texel >> 16) & 0xffu));
}

// Material::scatter (src/material.hpp:211-224) with the three material bodies (:72-86, :98-110,
// :127-156). Returns false when the path is absorbed.
// (PRE: `texel` was fetched by the caller with albedo_fetch where albedo_is_texture(m))
template <bool PRE = false>
RT_DEV bool scatter(const SceneDev& S, const MatRec& m, uint32_t& rng, f3 dir, f3 normal, float u, float v,
                    f3& out_dir, f3& out_att, uint32_t texel = 0) {
    // Diffuse (:72-86) and metallic (:98-110) share everything that is expensive — the three RNG draws + normalisation of
    // random_unit_vector and the albedo fetch — so they run as ONE instruction stream with the per-material operands selected:
    // a shading round of ~48 lanes nearly always holds both kinds, and two separate branches would execute that code twice.
    // Per lane the operations and their order are exactly those of its own material's body.
    if (m.type == RT_MAT_DIFFUSE || m.type == RT_MAT_METALLIC) {
        const bool diffuse = m.type == RT_MAT_DIFFUSE;
        const f3 unit = rng_unit_vector(rng);
        const f3 reflected = reflect3(dir, normal);                        // used by the metallic lanes only
        const f3 base = diffuse ? normal : reflected;
        const f3 offs = diffuse ? unit : m.roughness * unit;                // diffuse: normal + unit; metallic: reflected + roughness * unit
        out_dir = base + offs;
        if (diffuse && near_zero3(dir)) out_dir = normal;                   // tests the incoming direction, as the reference does
        if (PRE) out_att = m.tex_kind == RT_TEX_COLOR ? mk3(m.color[0], m.color[1], m.color[2]) : albedo_decode(texel);
        else out_att = albedo(S, m, u, v);
        return diffuse ? true : dot3(out_dir, normal) > 0.0f;
    }
    if (m.type == RT_MAT_DIELECTRIC) {
        out_att = mk3(1.0f, 1.0f, 1.0f);
        bool front_face = dot3(dir, normal) < 0.0f;
        f3 n = front_face ? normal : -normal;
        float ratio = front_face ? (1.0f / m.ior) : m.ior;
        f3 unit_direction = normalize3(dir);
        float cos_theta = __builtin_fminf(dot3(-unit_direction, n), 1.0f);
        float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
        bool reflect_it = ratio * sin_theta > 1.0f;
        if (!reflect_it) reflect_it = schlick(cos_theta, ratio) > rng_range(rng, 0.0f, 1.0f); // draw only if refractable
        out_dir = reflect_it ? reflect3(unit_direction, n) : refract3(unit_direction, n, ratio);
        return true;
    }
    out_dir = mk3(0.0f, 0.0f, 0.0f);
    out_att = mk3(0.0f, 0.0f, 0.0f);
    return false; // eNone
}

// ---- the scene's commonest normal matrices and materials in LDS (round 5) ------------------------------------------------------
// A shading round is a chain of dependent fetches — shading record -> material (+ instance) -> texel — run at the lower issue priority,
// and 14 % of a wave's time. The middle link reads two tiny tables (the atrium: 5 distinct normal matrices, 12 materials) through the
// vector memory path. Every workgroup of a kernel that shades stages the head of both tables in LDS (kLdsNm matrices, kLdsMats materials: 1,152 bytes): the material is
// then an LDS read away from the shading word, the texel fetch leaves one memory round trip after the shading record's, and the normal
// matrix is read where it is used instead of living in nine registers while the texel is in flight.
//   nm  : 3 x 16 bytes per matrix, the device's InstRec as it is (m[0..3], m[4..7], m[8], -, -, -)
//   mat : 2 x 16 bytes per material: {type | tex_kind << 8 | tex_layer << 16, colour rgb}, {emissive rgb, roughness (metallic) or ior (dielectric)}
// A round whose hits all index inside the staged part takes the LDS path; any other (a scene with more than kLdsNm matrices in use or
// more than kLdsMats materials, or without the packed shading word) takes the memory path — a wave-uniform choice per round.
struct ShadeTables {
    lds_u32x4* nm;
    lds_u32x4* mat;
    uint32_t n_nm, n_mat; // entries staged
};
RT_DEV void shade_tables_fill(const SceneDev& S, u32x4* nm, u32x4* mat) { // cooperative; the caller synchronises afterwards
    const u32x4* in = reinterpret_cast<const u32x4*>(S.inst);
    for (uint32_t i = threadIdx.x; i < S.lds_nm * 3u; i += blockDim.x) nm[i] = in[i];
    for (uint32_t i = threadIdx.x; i < S.lds_mats; i += blockDim.x) {
        const u32x4* mp = reinterpret_cast<const u32x4*>(S.mats + i);
        const u32x4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
        u32x4 a, b;
        a.x = m0.x | (m0.y << 8) | (m1.y << 16), a.y = m0.z, a.z = m0.w, a.w = m1.x;
        b.x = m1.z, b.y = m1.w, b.z = m2.x, b.w = m0.x == RT_MAT_DIELECTRIC ? m2.z : m2.y;
        mat[2u * i] = a, mat[2u * i + 1u] = b;
    }
}

// ---- trace_ray (src/trace_ray.hpp:11-82), the part after rtcIntersect1 ------------------------------
// Given the closest hit `h` of the ray (org, dir): returns true and sets `result` when the path ends
// (miss -> att*(sky+rad); absorbed -> att*rad); otherwise updates org/dir/att and returns false.
// `dir` is the unnormalised float direction.
// Diagnostic builds only (scripts/build_variant.sh NAME -DRT_ROUND_STAMPS): cycle stamps inside a shading round. `ck` points at eight wave-uniform
// accumulators (scalar registers), ck[7] = the previous stamp; RT_STAMP_AFTER adds the cycles since then to ck[i] once VAL is in its register
// (a stamp is a scalar memory round trip, ~300 cycles: the shares are what counts, not the sum).
#ifdef RT_ROUND_STAMPS
#define RT_STAMP_AFTER(ck, i, VAL)                                                     \
    do {                                                                              \
        if (ck) {                                                                     \
            asm volatile("" ::"v"(VAL));                                              \
            const long long n_ = (long long)__builtin_readcyclecounter();             \
            (ck)[i] += n_ - (ck)[7];                                                  \
            (ck)[7] = n_;                                                             \
        }                                                                             \
    } while (0)
#else
#define RT_STAMP_AFTER(ck, i, VAL) do { } while (0)
#endif
// USE_TAB = false: kernels that stage no tables (the dense SHADE kernel, the probes)
// `sink(colour)`: what the caller does with an ended path's contribution, called WHERE it is made (the miss branch, the absorbed branch). Returned
// through a reference it was live across the join of those branches — the point of a shading round where most is live — and the kernels that
// keep a ray per lane held it in scratch memory: three stores per path end and three reloads, each waited for on its own.
// (NoSink: the colour is returned in `result`. The instantiations are told apart by type, not by an `if`: hipcc's register allocation at the 80-register
// limit is sensitive to every extra value — the unsliced megakernel spills 2 dwords with `result`, 13 with a sink that stores into it.)
struct NoSink {};
template <class Sink> struct is_no_sink { static constexpr bool value = false; };
template <> struct is_no_sink<NoSink> { static constexpr bool value = true; };
template <bool USE_TAB = false, class Sink = NoSink>
RT_DEV bool shade_hit(const SceneDev& S, const ShadeTables& tab, uint32_t& rng, const Hit& h, f3& org, f3& dir, f3& att, f3& rad, f3& result,
                      long long* ck = nullptr, const Sink& sink = Sink{}) {
    if (h.tri == kNoTri) {
        if constexpr (is_no_sink<Sink>::value) result = att * (mk3(S.sky[0], S.sky[1], S.sky[2]) + rad);
        else sink(att * (mk3(S.sky[0], S.sky[1], S.sky[2]) + rad));
        return true;
    }
    const float4* sp = reinterpret_cast<const float4*>(S.shade + h.tri);
    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
    const f3 d = normalize3(dir);
    const f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
    const float uv0x = s2.y, uv0y = s2.z, uv1x = s2.w, uv1y = s3.x, uv2x = s3.y, uv2y = s3.z;
    const uint32_t iw = __float_as_uint(s3.w); // instance (| material << 20: rt_types.h)
    RT_STAMP_AFTER(ck, 1, iw); // the shading record has arrived
    // wave-uniform: every hit of this round finds its material and its normal matrix in the staged tables (n_nm = n_mat = 0 without the packed word)
    bool lds = false;
    if (USE_TAB) lds = __builtin_amdgcn_ballot_w64(!((iw & kPackedInstMask) < tab.n_nm && (iw >> kPackedInstBits) < tab.n_mat)) == 0ull;
    MatRec mat;
    if (USE_TAB && __builtin_expect(lds, 1)) {
        const uint32_t at = (iw >> kPackedInstBits) * 2u;
        const u32x4 a = tab.mat[at], b = tab.mat[at + 1u];
        mat.type = a.x & 0xffu, mat.tex_kind = (a.x >> 8) & 0xffu, mat.tex_layer = a.x >> 16;
        mat.color[0] = __uint_as_float(a.y), mat.color[1] = __uint_as_float(a.z), mat.color[2] = __uint_as_float(a.w);
        mat.emissive[0] = __uint_as_float(b.x), mat.emissive[1] = __uint_as_float(b.y), mat.emissive[2] = __uint_as_float(b.z);
        mat.roughness = mat.ior = __uint_as_float(b.w);
    } else {
        // The material record is read WHOLE, here, as three 16-byte words: one memory round trip after the shading record's, then one for
        // the texel. Read field by field at their uses (round 2) they became five more dependent fetches, each stuck behind a branch.
        const InstRec* inst = S.inst + (S.packed_mat ? (iw & kPackedInstMask) : iw);
        const u32x4* mp = reinterpret_cast<const u32x4*>(S.mats + (S.packed_mat ? (iw >> kPackedInstBits) : inst->material));
        const u32x4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
        mat.type = m0.x, mat.tex_kind = m0.y, mat.color[0] = __uint_as_float(m0.z), mat.color[1] = __uint_as_float(m0.w);
        mat.color[2] = __uint_as_float(m1.x), mat.tex_layer = m1.y, mat.emissive[0] = __uint_as_float(m1.z), mat.emissive[1] = __uint_as_float(m1.w);
        mat.emissive[2] = __uint_as_float(m2.x), mat.roughness = __uint_as_float(m2.y), mat.ior = __uint_as_float(m2.z);
    }
    const float bx = h.u, by = h.v;
    const float w = (1.0f - bx) - by;
    const float tu = (w * uv0x + bx * uv1x) + by * uv2x;
    const float tv = (w * uv0y + bx * uv1y) + by * uv2y;
    // The texel fetch, UNCONDITIONAL: a lane without a texture reads the first word of its own shading record instead (a line it has just
    // loaded) and ignores it. Under `if (textured)` the fetch sat in an exec-mask region of its own, and what hipcc would not move across
    // that region's ends — the normal matrix's loads among it — was issued after the texel's and waited for BEHIND it (vector memory
    // returns in order): the round paid the texel's latency with one normalisation to overlap it.
    uint32_t texel;
    {
        float fu = tu - __builtin_floorf(tu);
        float fv = tv - __builtin_floorf(tv);
        int iu = (int)__builtin_floorf(fu * 512.0f);
        int iv = (int)__builtin_floorf(fv * 512.0f);
        iu = iu < 0 ? 0 : (iu > 511 ? 511 : iu);
        iv = iv < 0 ? 0 : (iv > 511 ? 511 : iv);
        const uint8_t* ta = S.tex + (((size_t)mat.tex_layer * 512u + (size_t)iv) * 512u + (size_t)iu) * 4u;
        if (!albedo_is_texture(mat)) ta = reinterpret_cast<const uint8_t*>(sp);
        texel = *reinterpret_cast<const uint32_t*>(ta); // in flight during everything below
        RT_STAMP_AFTER(ck, 2, ta); // material read, uv interpolated, texel fetch issued
    }
    const bool emits = mat.type == RT_MAT_DIFFUSE || mat.type == RT_MAT_METALLIC; // Material::emitted :226-237 (here: its operands die early)
    rad = rad + (emits ? mk3(mat.emissive[0], mat.emissive[1], mat.emissive[2]) : mk3(0.0f, 0.0f, 0.0f));
    const f3 vn = normalize3((w * n0 + bx * n1) + by * n2);
    float nm[9]; // the normal matrix, read where it is used: nine registers less while the texel is in flight
    if (USE_TAB && __builtin_expect(lds, 1)) {
        const uint32_t at = (iw & kPackedInstMask) * 3u;
        const u32x4 a = tab.nm[at], b = tab.nm[at + 1u];
        const uint32_t c = *reinterpret_cast<__attribute__((address_space(3))) const uint32_t*>(tab.nm + at + 2u);
        nm[0] = __uint_as_float(a.x), nm[1] = __uint_as_float(a.y), nm[2] = __uint_as_float(a.z), nm[3] = __uint_as_float(a.w);
        nm[4] = __uint_as_float(b.x), nm[5] = __uint_as_float(b.y), nm[6] = __uint_as_float(b.z), nm[7] = __uint_as_float(b.w);
        nm[8] = __uint_as_float(c);
    } else {
        // (loads and their wait in ONE asm statement: hipcc's wait-count pass would otherwise merge this branch's pending loads into the
        // join below and make the LDS path wait for the TEXEL there — vmcnt counts in order — before it may touch the same registers)
        const InstRec* inst = S.inst + (S.packed_mat ? (iw & kPackedInstMask) : iw); // (computed again, here: as a pointer kept from the material fetch on it was 64 bits of scratch memory in the LDS path)
        u32x4 a, b;
        uint32_t c;
        asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                     "global_load_dwordx4 %1, %3, off offset:16\n\t"
                     "global_load_dword %2, %3, off offset:32\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c)
                     : "v"(inst)
                     : "memory");
        nm[0] = __uint_as_float(a.x), nm[1] = __uint_as_float(a.y), nm[2] = __uint_as_float(a.z), nm[3] = __uint_as_float(a.w);
        nm[4] = __uint_as_float(b.x), nm[5] = __uint_as_float(b.y), nm[6] = __uint_as_float(b.z), nm[7] = __uint_as_float(b.w);
        nm[8] = __uint_as_float(c);
    }
    const f3 g = mk3((nm[0] * vn.x + nm[3] * vn.y) + nm[6] * vn.z, (nm[1] * vn.x + nm[4] * vn.y) + nm[7] * vn.z,
                     (nm[2] * vn.x + nm[5] * vn.y) + nm[8] * vn.z);
    const f3 normal = normalize3(g);
    RT_STAMP_AFTER(ck, 3, normal.x); // normal interpolated, transformed, normalised twice
    f3 sdir, satt;
#ifdef RT_SCATTER_PRIO
    __builtin_amdgcn_s_setprio(RT_SCATTER_PRIO);
#endif
    const bool scattered = scatter<true>(S, mat, rng, d, normal, tu, tv, sdir, satt, texel);
    RT_STAMP_AFTER(ck, 4, satt.x); // scatter, with the texel's arrival
    if (scattered) {
        org = mk3(org.x + dir.x * h.t, org.y + dir.y * h.t, org.z + dir.z * h.t);
        dir = sdir;
        att = att * satt;
        return false;
    }
    if constexpr (is_no_sink<Sink>::value) result = att * rad;
    else sink(att * rad);
    return true;
}

// unorm8 image write (saturate, round to nearest even) followed by write_image's (uint8)(c*255)
// read-back (src/util.hpp:17-22)
RT_DEV uint8_t to_unorm8(float c) { return (uint8_t)__builtin_rintf(clamp01(c) * 255.0f); }

// LDS of a traversal kernel with BLOCK threads: the staged top of the BVH + the per-lane stacks
#define RT_TRAVERSAL_LDS(BLOCK)                                                                        \
    __shared__ u32x4 top_all[4 * kTopNodes]; /* the four planes back to back: trav_inner's fetch addresses them from ONE base */ \
    u32x4 *const top_w0 = top_all, *const top_w1 = top_all + kTopNodes, *const top_w2 = top_all + 2 * kTopNodes, *const top_ch = top_all + 3 * kTopNodes; \
    __shared__ int32_t lds_stack[kLdsStack * (BLOCK)];                                                 \
    int32_t spill[kStackSize - kLdsStack];                                                             \
    const int32_t top_count = (int32_t)S.n_nodes < kTopNodes ? (int32_t)S.n_nodes : kTopNodes;         \
    top_tree_fill(S, top_w0, top_w1, top_w2, top_ch, top_count);                                       \
    __syncthreads();                                                                                   \
    const TopTree top{(lds_u32x4*)top_w0, (lds_u32x4*)top_w1, (lds_u32x4*)top_w2, (lds_u32x4*)top_ch, top_count}; \
    const TravStack stack{(uint32_t)(size_t)(lds_i32*)(lds_stack + threadIdx.x), spill, 4u * (BLOCK)}; \
    lds_stack[threadIdx.x] = kTravDone; /* entry 0 of every lane's stack: the sentinel an empty stack pops (rt_device.h: trav_pop_lds) */

// ... and of a kernel that also shades: the head of the scene's normal-matrix and material tables (rt_device.h: ShadeTables). Before
// RT_TRAVERSAL_LDS, whose barrier covers this fill too.
#define RT_SHADE_LDS                                                                                   \
    __shared__ u32x4 shade_nm[3 * kLdsNm], shade_mat[2 * kLdsMats];                                    \
    shade_tables_fill(S, shade_nm, shade_mat);                                                         \
    const ShadeTables tab{(lds_u32x4*)shade_nm, (lds_u32x4*)shade_mat, S.lds_nm, S.lds_mats};


} // namespace rt
