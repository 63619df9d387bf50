// oracle_rt.cpp — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
//
// A scalar C++ restatement of the per-pixel ray-trace hot path of felipeagc/sycl-ray-tracer.
// It exists to CHECK the HIP product path (tests/, __graft_entry__.smoke(), bench.py's
// cpu_baseline leg). It is never linked into, imported by, or called from the product
// (sycl-ray-tracer_amd/): that path must fail loudly without its HIP library.
//
// PARITY UNPINNED: the reference has no tests, golden vectors or CPU code path, and its
// arithmetic below the shading code lives in un-vendored third parties (Embree 4, glm, the SYCL
// built-ins). This file therefore follows the reference SOURCE line by line
// (each function cites file:line under /root/reference) and pins what the source leaves open
// with the explicit rules below; it is anchored only by analytic known answers (tests/).
//
// Arithmetic rules (the "spec" the HIP kernels must reproduce bit for bit):
//   R1  every operation is a single IEEE-754 binary32 op, evaluated as written, left to right;
//       no FMA contraction (-ffp-contract=off), no fast-math, correctly rounded / and sqrt.
//   R2  dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z
//       cross(a,b) = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
//       normalize(v) = v * (1.0f / sqrtf(dot(v,v)))   (glm::normalize form; used for sycl's too)
//       length(v) = sqrtf(dot(v,v));  pow(x,5) = ((x*x)*(x*x))*x
//       clamp(x,0,1) = fminf(fmaxf(x,0),1)
//   R3  float -> half is round-to-nearest-even with IEEE overflow/subnormals; half -> float exact.
//   R4  std::hash<size_t> is the identity (libstdc++), so seeds are the pixel index itself.
//   R5  closest hit = over ALL triangles, Moller-Trumbore as written in mt_hit(); accept
//       t > 1e-4; minimum t wins, equal t -> lowest global triangle index. A BVH may only cull.
//       The triangle test is NOT the reference's arithmetic (that lives inside Embree: src/trace_ray.hpp:22) but this repository's
//       convention, so it is the one place where the operations are FUSED (round 4): its cross and dot products are
//       cross_f / dot_f below — every a*b + c of them a single fmaf (one rounding). R1 holds everywhere else.
//   R6  instance transform: world = ((m0*x + m4*y) + m8*z) + m12 per row (column-major 4x4).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fno-fast-math).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rt_mi355x.h" // ABI struct layouts only (no product code)

namespace {

// ------------------------------------------------------------------------------------------------
// vector math (R1, R2)
// ------------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) {
    float inv = 1.0f / sqrtf(dot(a, a));
    return mk(a.x * inv, a.y * inv, a.z * inv);
}
inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// ------------------------------------------------------------------------------------------------
// half precision storage (R3) — sycl::half conversions in RayData, src/camera.hpp:18-43
// ------------------------------------------------------------------------------------------------
inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { // inf / nan
        if (ax == 0x7f800000u) return (uint16_t)(sign | 0x7c00u);
        return (uint16_t)(sign | 0x7c00u | 0x200u | ((ax >> 13) & 0x3ffu));
    }
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); // >= 65520 rounds to inf
    if (ax < 0x38800000u) {                                   // below 2^-14: half subnormal / zero
        if (ax < 0x33000000u) return (uint16_t)sign;           // < 2^-25 rounds to zero
        const uint32_t e = ax >> 23;                           // 102..112
        const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
        const uint32_t shift = 126u - e; // 14..24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    const uint32_t e = (ax >> 23) - 112u;
    const uint32_t m = ax & 0x7fffffu;
    uint32_t r = (e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++; // carry may bump the exponent: correct
    return (uint16_t)(sign | r);
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    const uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else { // subnormal: m * 2^-24
            float v = (float)m * (1.0f / 16777216.0f);
            std::memcpy(&x, &v, 4);
            x |= sign;
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}
inline float hround(float f) { return f16_to_f32(f32_to_f16(f)); }

// ------------------------------------------------------------------------------------------------
// XorShift32State — src/xorshift.hpp:8-49
// ------------------------------------------------------------------------------------------------
struct Rng {
    uint32_t a;
    // operator()(): src/xorshift.hpp:11-20
    inline float next() {
        uint32_t x = a;
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        a = x;
        const float scale = 1.0f / 4294967296.0f; // 1.f / (uint64_t{1} << 32)
        return (float)a * scale;
    }
    // operator()(min,max): src/xorshift.hpp:22-24
    inline float next(float mn, float mx) { return mn + (mx - mn) * next(); }
    // vec(min,max): src/xorshift.hpp:30-36, argument evaluation pinned left to right (x,y,z)
    inline V3 vec(float mn, float mx) {
        float x = next(mn, mx);
        float y = next(mn, mx);
        float z = next(mn, mx);
        return mk(x, y, z);
    }
    // random_unit_vector: src/xorshift.hpp:38-40
    inline V3 unit_vector() { return normalize(vec(-1.0f, 1.0f)); }
};

// ------------------------------------------------------------------------------------------------
// scene: world-space triangles (R6) + shading attributes
// ------------------------------------------------------------------------------------------------
struct Tri {
    V3 v0, e1, e2;
};
struct Box {
    V3 lo, hi;
};
struct BNode { // oracle's own BVH (sweep SAH on one axis); independent of the product's builder
    Box box;
    int32_t left, right; // children, or left = -1 for a leaf
    uint32_t first, count;
};

struct Scene {
    std::vector<V3> normals;
    std::vector<float> uvs; // 2 per vertex
    std::vector<uint32_t> indices;
    std::vector<uint32_t> tri_instance;
    std::vector<rt_instance> instances;
    std::vector<rt_material> materials;
    std::vector<uint8_t> textures;
    uint32_t n_layers = 0;
    V3 sky{0.5f, 0.7f, 1.0f};
    std::vector<Tri> tris;     // global triangle order
    std::vector<V3> wverts;    // 3 per triangle, world space (for boxes)
    std::vector<BNode> nodes;  // BVH over tris
    std::vector<uint32_t> order; // leaf ranges index into this permutation
};

inline V3 xform_point(const float* m, V3 p) { // R6
    return mk(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12],
              ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
              ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}

struct Hit {
    float t, u, v;
    uint32_t tri;
};
constexpr uint32_t NO_TRI = 0xFFFFFFFFu;
constexpr float T_NEAR = 0.0001f; // RayData::to_embree tnear, src/camera.hpp:50

// R5: Moller-Trumbore; barycentrics follow Embree's convention P = (1-u-v) v0 + u v1 + v v2
// (usage at src/trace_ray.hpp:48-54). The fused forms of R5 (each fmaf rounds once):
inline float dot_f(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline V3 cross_f(V3 a, V3 b) {
    return mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
inline void mt_hit(const Tri& tr, uint32_t idx, V3 o, V3 d, Hit& best) {
    // Edge tests are made on the un-divided numerators (sign-corrected by det), so the one
    // correctly-rounded division is spent only on triangles that pass them.
    V3 p = cross_f(d, tr.e2);
    float det = dot_f(tr.e1, p);
    if (!(det != 0.0f)) return; // parallel (or NaN)
    V3 tv = o - tr.v0;
    V3 q = cross_f(tv, tr.e1);
    float sgn = det < 0.0f ? -1.0f : 1.0f;
    float un0 = dot_f(tv, p), vn0 = dot_f(d, q);
    float un = un0 * sgn;
    float vn = vn0 * sgn;
    if (!(un >= 0.0f && vn >= 0.0f && un + vn <= fabsf(det))) return;
    float inv = 1.0f / det;
    float t = dot_f(tr.e2, q) * inv;
    if (!(t > T_NEAR)) return;
    float u = un0 * inv;
    float v = vn0 * inv;
    if (t < best.t || (t == best.t && idx < best.tri)) {
        best.t = t;
        best.u = u;
        best.v = v;
        best.tri = idx;
    }
}

inline Hit intersect_brute(const Scene& sc, V3 o, V3 d) {
    Hit best{std::numeric_limits<float>::infinity(), 0.0f, 0.0f, NO_TRI};
    const uint32_t n = (uint32_t)sc.tris.size();
    for (uint32_t i = 0; i < n; ++i) mt_hit(sc.tris[i], i, o, d, best);
    return best;
}

// conservative slab test against a padded box (culling only; not part of the spec). Returns the entry
// distance, or a negative value when the box is missed.
inline float slab(const Box& b, V3 o, V3 inv, float tmax) {
    float t0 = (b.lo.x - o.x) * inv.x, t1 = (b.hi.x - o.x) * inv.x;
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    t0 = (b.lo.y - o.y) * inv.y;
    t1 = (b.hi.y - o.y) * inv.y;
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    t0 = (b.lo.z - o.z) * inv.z;
    t1 = (b.hi.z - o.z) * inv.z;
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    // boxes carry an absolute pad of 1e-4 x scene scale (build_bvh), orders of magnitude above the
    // few-ulp rounding of these products, so the plain comparison cannot cull a true overlap.
    // fminf/fmaxf drop the NaN of 0*inf (ray inside a slab plane with a zero direction component).
    tn = fmaxf(tn, 0.0f);
    return (tn <= tf && tn <= tmax) ? tn : -1.0f;
}

inline Hit intersect_bvh(const Scene& sc, V3 o, V3 d) {
    Hit best{std::numeric_limits<float>::infinity(), 0.0f, 0.0f, NO_TRI};
    if (sc.nodes.empty()) return best;
    V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    if (slab(sc.nodes[0].box, o, inv, best.t) < 0.0f) return best;
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const BNode& nd = sc.nodes[stack[--sp]];
        if (nd.left < 0) {
            for (uint32_t k = 0; k < nd.count; ++k) {
                uint32_t ti = sc.order[nd.first + k];
                mt_hit(sc.tris[ti], ti, o, d, best);
            }
            continue;
        }
        const float tl = slab(sc.nodes[nd.left].box, o, inv, best.t);
        const float tr = slab(sc.nodes[nd.right].box, o, inv, best.t);
        if (tl >= 0.0f && tr >= 0.0f) { // nearer child on top of the stack (visited first)
            if (tl <= tr) stack[sp++] = nd.right, stack[sp++] = nd.left;
            else stack[sp++] = nd.left, stack[sp++] = nd.right;
        } else if (tl >= 0.0f) {
            stack[sp++] = nd.left;
        } else if (tr >= 0.0f) {
            stack[sp++] = nd.right;
        }
    }
    return best;
}

void build_bvh(Scene& sc) {
    const uint32_t n = (uint32_t)sc.tris.size();
    sc.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) sc.order[i] = i;
    sc.nodes.clear();
    if (n == 0) return;
    // scene extent for the absolute pad
    V3 slo = sc.wverts[0], shi = sc.wverts[0];
    for (const V3& p : sc.wverts) {
        slo = mk(fminf(slo.x, p.x), fminf(slo.y, p.y), fminf(slo.z, p.z));
        shi = mk(fmaxf(shi.x, p.x), fmaxf(shi.y, p.y), fmaxf(shi.z, p.z));
    }
    const float ext = fmaxf(fmaxf(shi.x - slo.x, shi.y - slo.y), shi.z - slo.z);
    const float amax = fmaxf(fmaxf(fmaxf(fabsf(slo.x), fabsf(shi.x)), fmaxf(fabsf(slo.y), fabsf(shi.y))),
                             fmaxf(fabsf(slo.z), fabsf(shi.z)));
    const float pad = 1e-4f * fmaxf(ext, amax) + 1e-30f;
    std::vector<V3> cent(n);
    for (uint32_t i = 0; i < n; ++i) {
        cent[i] = (sc.wverts[3 * i] + sc.wverts[3 * i + 1] + sc.wverts[3 * i + 2]) * (1.0f / 3.0f);
    }
    struct Work {
        uint32_t node, first, count;
    };
    std::vector<Work> todo;
    sc.nodes.push_back(BNode{});
    todo.push_back({0, 0, n});
    while (!todo.empty()) {
        Work w = todo.back();
        todo.pop_back();
        V3 lo = mk(INFINITY, INFINITY, INFINITY), hi = mk(-INFINITY, -INFINITY, -INFINITY);
        V3 clo = lo, chi = hi;
        for (uint32_t k = 0; k < w.count; ++k) {
            uint32_t ti = sc.order[w.first + k];
            for (int j = 0; j < 3; ++j) {
                V3 p = sc.wverts[3 * ti + j];
                lo = mk(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z));
                hi = mk(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
            }
            V3 c = cent[ti];
            clo = mk(fminf(clo.x, c.x), fminf(clo.y, c.y), fminf(clo.z, c.z));
            chi = mk(fmaxf(chi.x, c.x), fmaxf(chi.y, c.y), fmaxf(chi.z, c.z));
        }
        BNode nd{};
        nd.box.lo = mk(lo.x - pad, lo.y - pad, lo.z - pad);
        nd.box.hi = mk(hi.x + pad, hi.y + pad, hi.z + pad);
        nd.first = w.first;
        nd.count = w.count;
        nd.left = nd.right = -1;
        if (w.count > 4) {
            // sweep SAH along the axis of largest centroid extent (the oracle's own builder; the image does
            // not depend on it). Falls back to the median when every candidate is degenerate.
            V3 e = chi - clo;
            int axis = (e.x >= e.y && e.x >= e.z) ? 0 : (e.y >= e.z ? 1 : 2);
            auto key = [&](uint32_t ti) {
                return axis == 0 ? cent[ti].x : (axis == 1 ? cent[ti].y : cent[ti].z);
            };
            std::sort(sc.order.begin() + w.first, sc.order.begin() + w.first + w.count,
                      [&](uint32_t a, uint32_t b) { return key(a) < key(b) || (key(a) == key(b) && a < b); });
            auto area = [](V3 l, V3 h) {
                V3 d = h - l;
                return d.x * d.y + d.y * d.z + d.z * d.x;
            };
            std::vector<float> right_area(w.count);
            V3 bl = mk(INFINITY, INFINITY, INFINITY), bh = mk(-INFINITY, -INFINITY, -INFINITY);
            for (uint32_t k = w.count; k-- > 0;) {
                uint32_t ti = sc.order[w.first + k];
                for (int j = 0; j < 3; ++j) {
                    V3 p = sc.wverts[3 * ti + j];
                    bl = mk(fminf(bl.x, p.x), fminf(bl.y, p.y), fminf(bl.z, p.z));
                    bh = mk(fmaxf(bh.x, p.x), fmaxf(bh.y, p.y), fmaxf(bh.z, p.z));
                }
                right_area[k] = area(bl, bh);
            }
            bl = mk(INFINITY, INFINITY, INFINITY), bh = mk(-INFINITY, -INFINITY, -INFINITY);
            uint32_t mid = w.count / 2;
            float best_cost = INFINITY;
            for (uint32_t k = 0; k + 1 < w.count; ++k) {
                uint32_t ti = sc.order[w.first + k];
                for (int j = 0; j < 3; ++j) {
                    V3 p = sc.wverts[3 * ti + j];
                    bl = mk(fminf(bl.x, p.x), fminf(bl.y, p.y), fminf(bl.z, p.z));
                    bh = mk(fmaxf(bh.x, p.x), fmaxf(bh.y, p.y), fmaxf(bh.z, p.z));
                }
                float cost = area(bl, bh) * (float)(k + 1) + right_area[k + 1] * (float)(w.count - k - 1);
                if (cost < best_cost) best_cost = cost, mid = k + 1;
            }
            nd.left = (int32_t)sc.nodes.size();
            nd.right = nd.left + 1;
            sc.nodes.push_back(BNode{});
            sc.nodes.push_back(BNode{});
            todo.push_back({(uint32_t)nd.left, w.first, mid});
            todo.push_back({(uint32_t)nd.right, w.first + mid, w.count - mid});
        }
        sc.nodes[w.node] = nd;
    }
}

// ------------------------------------------------------------------------------------------------
// camera — src/camera.hpp:74-131
// ------------------------------------------------------------------------------------------------
// Camera::Camera: src/camera.hpp:74-106 (host-side sycl::float3 math)
void camera_init(rt_camera* c, int32_t w, int32_t h, V3 center, V3 cam_dir, float focal) {
    c->width = w;
    c->height = h;
    V3 dir = normalize(cam_dir);
    V3 world_up = mk(0.0f, 1.0f, 0.0f);
    V3 right = normalize(cross(dir, world_up));
    V3 up = normalize(cross(right, dir));
    float vp0 = 1.0f * ((float)w / (float)h);
    float vp1 = 1.0f;
    V3 viewport_u = (-right) * vp0;
    V3 viewport_v = up * vp1;
    V3 p00 = ((center + viewport_u) + viewport_v) + dir * focal;
    V3 du = right / ((float)w / (vp0 * 2.0f));
    V3 dv = (-up) / ((float)h / (vp1 * 2.0f));
    c->center[0] = center.x, c->center[1] = center.y, c->center[2] = center.z;
    c->pixel00[0] = p00.x, c->pixel00[1] = p00.y, c->pixel00[2] = p00.z;
    c->delta_u[0] = du.x, c->delta_u[1] = du.y, c->delta_u[2] = du.z;
    c->delta_v[0] = dv.x, c->delta_v[1] = dv.y, c->delta_v[2] = dv.z;
}

// the live state of one path: RayData (src/camera.hpp:12-44): org f32, dir/att/rad stored as half
struct Ray {
    V3 org;
    uint16_t dir[3], att[3], rad[3];
};
inline void store3(uint16_t* h, V3 v) {
    h[0] = f32_to_f16(v.x);
    h[1] = f32_to_f16(v.y);
    h[2] = f32_to_f16(v.z);
}
inline V3 load3(const uint16_t* h) { return mk(f16_to_f32(h[0]), f16_to_f32(h[1]), f16_to_f32(h[2])); }

// Camera::get_ray + pixel_sample_square: src/camera.hpp:109-131; RayData ctor :30-44
inline Ray get_ray(const rt_camera& c, int x, int y, Rng& rng) {
    V3 p00 = mk(c.pixel00[0], c.pixel00[1], c.pixel00[2]);
    V3 du = mk(c.delta_u[0], c.delta_u[1], c.delta_u[2]);
    V3 dv = mk(c.delta_v[0], c.delta_v[1], c.delta_v[2]);
    V3 center = mk(c.center[0], c.center[1], c.center[2]);
    V3 pixel_center = (p00 + ((float)x * du)) + ((float)y * dv);
    float px = -0.5f + rng.next();
    float py = -0.5f + rng.next();
    V3 sq = (px * du) + (py * dv);
    V3 pixel_sample = pixel_center + sq;
    V3 direction = pixel_sample - center;
    Ray r;
    r.org = center;
    store3(r.dir, direction);
    store3(r.att, mk(1.0f, 1.0f, 1.0f));
    store3(r.rad, mk(0.0f, 0.0f, 0.0f));
    return r;
}

// ------------------------------------------------------------------------------------------------
// materials — src/material.hpp, src/util.hpp:82-125
// ------------------------------------------------------------------------------------------------
// near_zero: src/util.hpp:103-107
inline bool near_zero(V3 e) {
    const float s = 1e-8f;
    return (fabsf(e.x) < s) && (fabsf(e.y) < s) && (fabsf(e.z) < s);
}
// length_squared: src/util.hpp:109-112 — literally length(v) * length(v)
inline float length_squared(V3 v) {
    float l = length(v);
    return l * l;
}
// reflect: src/util.hpp:114-116
inline V3 reflect(V3 v, V3 n) { return v - (2.0f * dot(v, n)) * n; }
// refract: src/util.hpp:118-125
inline V3 refract(V3 uv, V3 n, float etai_over_etat) {
    float cos_theta = fminf(dot(-uv, n), 1.0f);
    V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V3 r_out_parallel = (-sqrtf(fabsf(1.0f - length_squared(r_out_perp)))) * n;
    return r_out_perp + r_out_parallel;
}

// Texture::sample: src/material.hpp:45-53; sampler = normalized coords, repeat, nearest
// (src/render_wavefront.cpp:193-197); unorm8 texel -> c/255.
inline V3 sample_albedo(const Scene& sc, const rt_material& m, float u, float v) {
    if (m.tex_kind == RT_TEX_COLOR) return mk(m.color[0], m.color[1], m.color[2]);
    float fu = u - floorf(u);
    float fv = v - floorf(v);
    int iu = (int)floorf(fu * 512.0f);
    int iv = (int)floorf(fv * 512.0f);
    iu = iu < 0 ? 0 : (iu > 511 ? 511 : iu);
    iv = iv < 0 ? 0 : (iv > 511 ? 511 : iv);
    const uint8_t* px =
        &sc.textures[(((size_t)m.tex_layer * 512u + (size_t)iv) * 512u + (size_t)iu) * 4u];
    return mk((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

struct Scatter {
    V3 dir, att;
};

// MaterialDielectric::reflectance: src/material.hpp:120-125 (pow(x,5) per R2)
inline float reflectance(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x5 = (x2 * x2) * x;
    return r0 + (1.0f - r0) * x5;
}

// Material::scatter dispatch: src/material.hpp:211-224
inline bool scatter(const Scene& sc, const rt_material& m, Rng& rng, V3 dir, V3 normal, float u,
                    float v, Scatter& out) {
    switch (m.type) {
    case RT_MAT_DIFFUSE: { // MaterialDiffuse::scatter: src/material.hpp:72-86
        out.dir = normal + rng.unit_vector();
        if (near_zero(dir)) out.dir = normal; // tests the INCOMING dir (kept as in the reference)
        out.att = sample_albedo(sc, m, u, v);
        return true;
    }
    case RT_MAT_METALLIC: { // MaterialMetallic::scatter: src/material.hpp:98-110
        V3 reflected = reflect(dir, normal);
        out.dir = reflected + m.roughness * rng.unit_vector();
        out.att = sample_albedo(sc, m, u, v);
        return dot(out.dir, normal) > 0.0f;
    }
    case RT_MAT_DIELECTRIC: { // MaterialDielectric::scatter: src/material.hpp:127-156
        out.att = mk(1.0f, 1.0f, 1.0f);
        bool front_face = dot(dir, normal) < 0.0f;
        V3 n = front_face ? normal : -normal;
        float ratio = front_face ? (1.0f / m.ior) : m.ior;
        V3 unit_direction = normalize(dir);
        float cos_theta = fminf(dot(-unit_direction, n), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        bool cannot_refract = ratio * sin_theta > 1.0f;
        // short-circuit ||: the RNG draw happens only when refraction is possible
        if (cannot_refract || reflectance(cos_theta, ratio) > rng.next(0.0f, 1.0f)) {
            out.dir = reflect(unit_direction, n);
        } else {
            out.dir = refract(unit_direction, n, ratio);
        }
        return true;
    }
    default: return false; // eNone
    }
}

// Material::emitted: src/material.hpp:226-237
inline V3 emitted(const rt_material& m) {
    if (m.type == RT_MAT_DIFFUSE || m.type == RT_MAT_METALLIC)
        return mk(m.emissive[0], m.emissive[1], m.emissive[2]);
    return mk(0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------------
// trace_ray — src/trace_ray.hpp:11-82.  Returns true and sets `result` when the path ends.
// org/dir are the float ray (dir already widened from half); on a scatter they are updated.
// ------------------------------------------------------------------------------------------------
inline bool trace_ray(const Scene& sc, bool use_bvh, Rng& rng, V3& org, V3& dir, V3& att, V3& rad,
                      V3& result) {
    Hit h = use_bvh ? intersect_bvh(sc, org, dir) : intersect_brute(sc, org, dir);
    if (h.tri == NO_TRI) { // :25-27
        result = att * (sc.sky + rad);
        return true;
    }
    const uint32_t inst_id = sc.tri_instance[h.tri];
    const rt_instance& inst = sc.instances[inst_id]; // GeometryData, :29-30
    const rt_material& mat = sc.materials[inst.material];
    const uint32_t i0 = sc.indices[3 * h.tri], i1 = sc.indices[3 * h.tri + 1],
                   i2 = sc.indices[3 * h.tri + 2];
    const float bx = h.u, by = h.v;
    const float w = (1.0f - bx) - by;
    // uv: :48-49
    const float tu = (w * sc.uvs[2 * i0] + bx * sc.uvs[2 * i1]) + by * sc.uvs[2 * i2];
    const float tv = (w * sc.uvs[2 * i0 + 1] + bx * sc.uvs[2 * i1 + 1]) + by * sc.uvs[2 * i2 + 1];
    // normal: :52-59
    V3 vn = normalize((w * sc.normals[i0] + bx * sc.normals[i1]) + by * sc.normals[i2]);
    const float* nm = inst.normal_mat; // column-major mat3 * vec3 (glm)
    V3 g = mk((nm[0] * vn.x + nm[3] * vn.y) + nm[6] * vn.z, (nm[1] * vn.x + nm[4] * vn.y) + nm[7] * vn.z,
              (nm[2] * vn.x + nm[5] * vn.y) + nm[8] * vn.z);
    V3 normal = normalize(g);
    V3 d = normalize(dir); // :61-62
    rad = rad + emitted(mat); // :64
    Scatter s;
    if (scatter(sc, mat, rng, d, normal, tu, tv, s)) { // :67-76
        org = mk(org.x + dir.x * h.t, org.y + dir.y * h.t, org.z + dir.z * h.t);
        dir = s.dir;
        att = att * s.att;
        return false;
    }
    result = att * rad; // :77-79
    return true;
}

inline uint32_t seed_for(int x, int y, int w, int h, uint32_t seed_mode) {
    if (seed_mode == RT_SEED_MEGAKERNEL) { // nd_range<2> global linear id, src/render_megakernel.cpp:90-93,144-146
        uint32_t h8 = (uint32_t)((h + 7) / 8) * 8u;
        return (uint32_t)x * h8 + (uint32_t)y;
    }
    return (uint32_t)x + (uint32_t)y * (uint32_t)w; // src/render_wavefront.cpp:69-73
}

// Russian roulette — an EXTENSION (the reference only lists it as a to-do: PLAN.md:23-27), off unless
// rr_start > 0, never part of a parity run against the reference's own numbers. Applied to a path that
// continues after its bounce number b (1-based) when rr_start <= b < max_depth:
//   q = the attenuation as it is stored (three halves);  p = min(max(max(q.r,q.g,q.b), 0.05), 1);
//   one draw u;  the path survives iff u < p and carries on with q / p (three fp32 divisions),
//   otherwise it ends and contributes (0,0,0) like a path that outlives max_depth.
inline bool roulette(Rng& rng, V3& att) {
    uint16_t h[3];
    store3(h, att);
    const V3 q = load3(h);
    const float p = fminf(fmaxf(fmaxf(q.x, fmaxf(q.y, q.z)), 0.05f), 1.0f);
    const float u = rng.next();
    if (!(u < p)) return false;
    att = mk(q.x / p, q.y / p, q.z / p);
    return true;
}

// render_pixel: src/render_megakernel.cpp:20-63
inline V3 render_pixel(const Scene& sc, bool use_bvh, const rt_camera& cam, Rng& rng, int x, int y,
                       uint32_t max_depth, uint32_t rr_start, uint64_t& ray_count) {
    Ray rd = get_ray(cam, x, y, rng);
    for (uint32_t i = 0; i < max_depth; ++i) {
        ray_count++;
        V3 att = load3(rd.att);
        V3 rad = load3(rd.rad);
        V3 org = rd.org;
        V3 dir = load3(rd.dir);
        V3 res;
        bool done = trace_ray(sc, use_bvh, rng, org, dir, att, rad, res);
        if (!done && rr_start && i + 1 >= rr_start && i + 1 < max_depth && !roulette(rng, att)) return mk(0.0f, 0.0f, 0.0f);
        rd.org = org;
        store3(rd.dir, dir);
        store3(rd.att, att);
        store3(rd.rad, rad);
        if (done) return res;
    }
    return mk(0.0f, 0.0f, 0.0f);
}

inline uint8_t to_unorm8(float c) { // unorm8 image write (saturate, round to nearest even) then
    float s = clamp01(c) * 255.0f;  // write_image's (uint8)(c*255) read-back: src/util.hpp:17-22
    return (uint8_t)lrintf(s);
}

} // namespace

// ================================================================================================
// C entry points (ctypes)
// ================================================================================================
extern "C" {

struct orc_scene {
    Scene sc;
};

int orc_scene_create(const rt_scene_desc* d, orc_scene** out) {
    if (!d || !out) return -1;
    for (uint32_t t = 0; t < d->n_triangles; ++t) {
        if (d->tri_instance[t] >= d->n_instances) return -1;
        for (int k = 0; k < 3; ++k)
            if (d->indices[3 * t + k] >= d->n_vertices) return -1;
    }
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        if (d->instances[i].material >= d->n_materials) return -1;
    }
    orc_scene* h = new orc_scene();
    Scene& sc = h->sc;
    sc.normals.resize(d->n_vertices);
    for (uint32_t i = 0; i < d->n_vertices; ++i)
        sc.normals[i] = mk(d->normals[3 * i], d->normals[3 * i + 1], d->normals[3 * i + 2]);
    sc.uvs.assign(d->uvs, d->uvs + 2 * (size_t)d->n_vertices);
    sc.indices.assign(d->indices, d->indices + 3 * (size_t)d->n_triangles);
    sc.tri_instance.assign(d->tri_instance, d->tri_instance + d->n_triangles);
    sc.instances.assign(d->instances, d->instances + d->n_instances);
    sc.materials.assign(d->materials, d->materials + d->n_materials);
    sc.n_layers = d->n_layers;
    if (d->n_layers) sc.textures.assign(d->textures, d->textures + (size_t)d->n_layers * 512u * 512u * 4u);
    sc.sky = mk(d->sky[0], d->sky[1], d->sky[2]);
    sc.tris.resize(d->n_triangles);
    sc.wverts.resize(3 * (size_t)d->n_triangles);
    for (uint32_t t = 0; t < d->n_triangles; ++t) {
        const float* m = d->instances[d->tri_instance[t]].transform;
        V3 p[3];
        for (int k = 0; k < 3; ++k) {
            uint32_t vi = d->indices[3 * t + k];
            p[k] = xform_point(m, mk(d->positions[3 * vi], d->positions[3 * vi + 1], d->positions[3 * vi + 2]));
            sc.wverts[3 * (size_t)t + k] = p[k];
        }
        sc.tris[t] = Tri{p[0], p[1] - p[0], p[2] - p[0]};
    }
    build_bvh(sc);
    *out = h;
    return 0;
}
void orc_scene_destroy(orc_scene* s) { delete s; }

int orc_camera_init(rt_camera* out, int32_t w, int32_t h, const float* center, const float* dir,
                    float focal) {
    camera_init(out, w, h, mk(center[0], center[1], center[2]), mk(dir[0], dir[1], dir[2]), focal);
    return 0;
}

void orc_xorshift(uint32_t seed, uint32_t n, float* out, uint32_t* state_out) {
    Rng r{seed};
    for (uint32_t i = 0; i < n; ++i) out[i] = r.next();
    if (state_out) *state_out = r.a;
}

void orc_half_roundtrip(uint32_t n, const float* in, float* out, uint16_t* bits) {
    for (uint32_t i = 0; i < n; ++i) {
        uint16_t h = f32_to_f16(in[i]);
        if (bits) bits[i] = h;
        out[i] = f16_to_f32(h);
    }
}

int orc_intersect(const orc_scene* s, uint32_t n, const float* org, const float* dir, int use_bvh,
                  float* t, float* u, float* v, uint32_t* tri) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        V3 d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        Hit h = use_bvh ? intersect_bvh(s->sc, o, d) : intersect_brute(s->sc, o, d);
        t[i] = h.t;
        u[i] = h.u;
        v[i] = h.v;
        tri[i] = h.tri;
    }
    return 0;
}

int orc_scatter(const orc_scene* s, uint32_t material, uint32_t n, const float* dir,
                const float* normal, const float* uv, const uint32_t* seed, uint8_t* ok,
                float* out_dir, float* out_att, uint32_t* seed_out) {
    if (material >= s->sc.materials.size()) return -1;
    const rt_material& m = s->sc.materials[material];
    for (uint32_t i = 0; i < n; ++i) {
        Rng rng{seed[i]};
        Scatter sr{mk(0, 0, 0), mk(0, 0, 0)};
        bool r = scatter(s->sc, m, rng, mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                         mk(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]), uv[2 * i], uv[2 * i + 1], sr);
        ok[i] = r ? 1 : 0;
        out_dir[3 * i] = sr.dir.x, out_dir[3 * i + 1] = sr.dir.y, out_dir[3 * i + 2] = sr.dir.z;
        out_att[3 * i] = sr.att.x, out_att[3 * i + 1] = sr.att.y, out_att[3 * i + 2] = sr.att.z;
        seed_out[i] = rng.a;
    }
    return 0;
}

// Renders global rows [row_begin,row_end) selected by the strip rule (row/strip_rows) % world == rank
// into a compact top-to-bottom buffer, exactly like the product's tile split.
// kind: RT_RENDERER_MEGAKERNEL restates K1 (src/render_megakernel.cpp:142-158);
//       RT_RENDERER_WAVEFRONT restates K2-K6 (src/render_wavefront.cpp:62-74,106-124,244-293,
//       340-354,377-390,396-417) with per-sample queues.
int orc_render(const orc_scene* s, const rt_camera* cam, int kind, uint32_t seed_mode,
               uint32_t max_depth, uint32_t spp, int use_bvh, int nthreads, uint32_t rank,
               uint32_t world, uint32_t strip_rows, uint32_t rr_start, float* rgba_f32, uint8_t* rgba_u8,
               uint64_t* rays_out) {
    if (!s || !cam || world == 0 || rank >= world || strip_rows == 0) return -1;
    const Scene& sc = s->sc;
    const int W = cam->width, H = cam->height;
    if (seed_mode == RT_SEED_DEFAULT)
        seed_mode = kind == RT_RENDERER_MEGAKERNEL ? RT_SEED_MEGAKERNEL : RT_SEED_WAVEFRONT;
    std::vector<int> rows;
    for (int y = 0; y < H; ++y)
        if (((uint32_t)y / strip_rows) % world == rank) rows.push_back(y);
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    std::atomic<uint64_t> total{0};
    std::atomic<size_t> next_row{0};
    const bool bvh = use_bvh != 0;

    auto emit = [&](size_t lr, int x, V3 mean_sqrt) {
        size_t o = (lr * (size_t)W + (size_t)x) * 4;
        if (rgba_f32) {
            rgba_f32[o] = mean_sqrt.x, rgba_f32[o + 1] = mean_sqrt.y, rgba_f32[o + 2] = mean_sqrt.z;
            rgba_f32[o + 3] = 1.0f;
        }
        if (rgba_u8) {
            rgba_u8[o] = to_unorm8(mean_sqrt.x), rgba_u8[o + 1] = to_unorm8(mean_sqrt.y);
            rgba_u8[o + 2] = to_unorm8(mean_sqrt.z), rgba_u8[o + 3] = 255;
        }
    };

    auto worker_mega = [&]() {
        uint64_t rays = 0;
        for (;;) {
            size_t lr = next_row.fetch_add(1);
            if (lr >= rows.size()) break;
            int y = rows[lr];
            for (int x = 0; x < W; ++x) {
                Rng rng{seed_for(x, y, W, H, seed_mode)};
                V3 color = mk(0, 0, 0);
                for (uint32_t i = 0; i < spp; ++i)
                    color = color + render_pixel(sc, bvh, *cam, rng, x, y, max_depth, rr_start, rays);
                color = color / (float)spp;                             // :154
                emit(lr, x, mk(sqrtf(color.x), sqrtf(color.y), sqrtf(color.z))); // linear_to_gamma :156
            }
        }
        total += rays;
    };

    // Wavefront restatement: one row at a time is a valid partition because rays of different
    // pixels never interact; queues are explicit ping-pong arrays as in Buffers
    // (src/render_wavefront.hpp:10-38).
    auto worker_wave = [&]() {
        uint64_t rays = 0;
        std::vector<uint32_t> rngs(W);
        std::vector<float> image(4 * (size_t)W), combined(4 * (size_t)W);
        struct Q {
            std::vector<uint32_t> id;
            std::vector<V3> org;
            std::vector<uint16_t> dir, att, rad;
            size_t len = 0;
        } q[2];
        for (auto& b : q) {
            b.id.resize(W), b.org.resize(W), b.dir.resize(3 * (size_t)W), b.att.resize(3 * (size_t)W);
            b.rad.resize(3 * (size_t)W);
        }
        for (;;) {
            size_t lr = next_row.fetch_add(1);
            if (lr >= rows.size()) break;
            int y = rows[lr];
            for (int x = 0; x < W; ++x) rngs[x] = seed_for(x, y, W, H, seed_mode); // K2 :69-73
            std::fill(combined.begin(), combined.end(), 0.0f);
            uint32_t buffer_index = 0;
            for (uint32_t sample = 0; sample < spp; ++sample) {
                // K3 generate_camera_rays :106-124
                Q& g = q[buffer_index & 1];
                for (int x = 0; x < W; ++x) {
                    image[4 * x] = image[4 * x + 1] = image[4 * x + 2] = image[4 * x + 3] = 0.0f;
                    Rng rng{rngs[x]};
                    Ray r = get_ray(*cam, x, y, rng);
                    rngs[x] = rng.a;
                    g.id[x] = (uint32_t)x;
                    g.org[x] = r.org;
                    std::memcpy(&g.dir[3 * x], r.dir, 6), std::memcpy(&g.att[3 * x], r.att, 6);
                    std::memcpy(&g.rad[3 * x], r.rad, 6);
                }
                g.len = (size_t)W;
                for (uint32_t depth = 0; depth < max_depth; ++depth) { // :406-412
                    rays += q[buffer_index & 1].len;
                    buffer_index++;
                    Q& prev = q[~buffer_index & 1];
                    Q& cur = q[buffer_index & 1];
                    cur.len = 0;
                    // K4 shoot_rays :244-311
                    for (size_t gi = 0; gi < prev.len; ++gi) {
                        uint32_t id = prev.id[gi];
                        V3 org = prev.org[gi];
                        V3 dir = load3(&prev.dir[3 * gi]);
                        V3 att = load3(&prev.att[3 * gi]);
                        V3 rad = load3(&prev.rad[3 * gi]);
                        Rng rng{rngs[id]};
                        V3 res;
                        bool done = trace_ray(sc, bvh, rng, org, dir, att, rad, res);
                        bool killed = false;
                        if (!done && rr_start && depth + 1 >= rr_start && depth + 1 < max_depth) killed = !roulette(rng, att);
                        rngs[id] = rng.a;
                        if (done) { // :275-278
                            image[4 * id] = clamp01(res.x), image[4 * id + 1] = clamp01(res.y);
                            image[4 * id + 2] = clamp01(res.z), image[4 * id + 3] = 1.0f;
                        } else if (depth == max_depth - 1 || killed) { // :279-280
                            image[4 * id] = image[4 * id + 1] = image[4 * id + 2] = 0.0f;
                            image[4 * id + 3] = 1.0f;
                        } else { // :281-292
                            size_t k = cur.len++;
                            cur.id[k] = id;
                            cur.org[k] = org;
                            store3(&cur.dir[3 * k], dir), store3(&cur.att[3 * k], att);
                            store3(&cur.rad[3 * k], rad);
                        }
                    }
                    prev.len = 0;
                }
                // K5 merge_samples :350-353
                for (size_t k = 0; k < 4 * (size_t)W; ++k) combined[k] = combined[k] + image[k];
            }
            // K6 convert_image_to_srgb :386-389
            for (int x = 0; x < W; ++x) {
                V3 c = mk(combined[4 * x] / (float)spp, combined[4 * x + 1] / (float)spp,
                          combined[4 * x + 2] / (float)spp);
                emit(lr, x, mk(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z)));
            }
        }
        total += rays;
    };

    std::vector<std::thread> th;
    for (int i = 0; i < nthreads; ++i) {
        if (kind == RT_RENDERER_MEGAKERNEL)
            th.emplace_back(worker_mega);
        else
            th.emplace_back(worker_wave);
    }
    for (auto& t : th) t.join();
    if (rays_out) *rays_out = total.load();
    return 0;
}

int orc_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

} // extern "C"
