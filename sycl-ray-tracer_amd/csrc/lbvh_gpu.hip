// lbvh_gpu.hip — BVH construction on the GPU (RT_BVH_LBVH_GPU): Morton codes, device radix sort,
// Karras' binary radix tree, bottom-up refit, breadth-first collapse to the quantised BVH4 the traversal
// kernels use. It replaces the host builder when scene (re)build time matters (the
// reference rebuilds through Embree's rtcCommitScene, src/scene.cpp:101-107). The image does not depend on
// the tree, so this builder is validated structurally (rt_scene_check_bvh on the downloaded tree) and by
// the same render parity tests as the host builders.
//
// Pipeline (all on one stream):
//   k_prims    world-space triangle boxes + 63-bit Morton code of the box centre, key = code
//   rocPRIM    radix sort of (key, triangle index) pairs
//   k_karras   one thread per internal node of the binary radix tree (keys made unique by the index)
//   k_refit    one thread per leaf walks up; the second arrival at a node merges its children's boxes
//   k_emit     one launch per BVH4 level: a node takes its binary node's grandchildren as its (<= 4) children,
//              quantises their padded boxes conservatively and appends inner children to the next frontier,
//              so node indices are breadth-first (the first kTopNodes nodes are the shallowest: LDS staging)
//   k_permute  triangles into leaf (= Morton) order
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <string>
#include <vector>

#include "scene_build.h"

namespace rt {
namespace {

#define LB_CHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            return e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP;                \
        }                                                                              \
    } while (0)

struct Box3 {
    float lo[3], hi[3];
};

__device__ __forceinline__ unsigned long long expand21(unsigned long long v) {
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

__global__ void __launch_bounds__(256) k_prims(uint32_t n, const float* __restrict__ wverts, float3 clo, float3 cext_inv,
                                                Box3* __restrict__ tri_box, unsigned long long* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const float* w = wverts + 9 * (size_t)t;
    Box3 b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(w[a], fminf(w[3 + a], w[6 + a]));
        b.hi[a] = fmaxf(w[a], fmaxf(w[3 + a], w[6 + a]));
    }
    tri_box[t] = b;
    const float c[3] = {0.5f * (b.lo[0] + b.hi[0]), 0.5f * (b.lo[1] + b.hi[1]), 0.5f * (b.lo[2] + b.hi[2])};
    const float ci[3] = {cext_inv.x, cext_inv.y, cext_inv.z}, cl[3] = {clo.x, clo.y, clo.z};
    unsigned long long code = 0;
    for (int a = 0; a < 3; ++a) {
        float f = (c[a] - cl[a]) * ci[a];
        f = fminf(fmaxf(f, 0.0f), 1.0f);
        unsigned long long q = (unsigned long long)fminf(f * 2097152.0f, 2097151.0f);
        code |= expand21(q) << (2 - a);
    }
    keys[t] = code;
    vals[t] = t;
}

// common-prefix length of the (key, position) pairs i and j; positions break ties so every pair is unique
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = k[i], b = k[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// child reference of the binary tree: >= 0 internal node, < 0 leaf ~position
__global__ void __launch_bounds__(256) k_karras(int n, const unsigned long long* __restrict__ keys, int* __restrict__ left,
                                                 int* __restrict__ right, int* __restrict__ parent_inner,
                                                 int* __restrict__ parent_leaf) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int L = lo == gamma ? ~gamma : gamma;
    const int R = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    left[i] = L, right[i] = R;
    if (L >= 0) parent_inner[L] = i; else parent_leaf[~L] = i;
    if (R >= 0) parent_inner[R] = i; else parent_leaf[~R] = i;
    if (i == 0) parent_inner[0] = -1;
}

__global__ void __launch_bounds__(256) k_refit(int n, const uint32_t* __restrict__ sorted_tri, const Box3* __restrict__ tri_box,
                                                const int* __restrict__ left, const int* __restrict__ right,
                                                const int* __restrict__ parent_inner, const int* __restrict__ parent_leaf,
                                                Box3* __restrict__ node_box, unsigned int* __restrict__ flags) {
    const int p = (int)(blockIdx.x * 256u + threadIdx.x);
    if (p >= n) return;
    int node = parent_leaf[p];
    while (node >= 0) {
        __threadfence();                              // release: our subtree's boxes are visible before the arrival
        if (atomicAdd(&flags[node], 1u) == 0u) return; // first arrival: the sibling subtree is not finished yet
        __threadfence();                              // acquire: see the sibling's boxes
        Box3 b;
        const int c[2] = {left[node], right[node]};
        for (int a = 0; a < 3; ++a) b.lo[a] = INFINITY, b.hi[a] = -INFINITY;
        for (int k = 0; k < 2; ++k) {
            const Box3 cb = c[k] >= 0 ? node_box[c[k]] : tri_box[sorted_tri[~c[k]]];
            for (int a = 0; a < 3; ++a) b.lo[a] = fminf(b.lo[a], cb.lo[a]), b.hi[a] = fmaxf(b.hi[a], cb.hi[a]);
        }
        node_box[node] = b;
        node = parent_inner[node];
    }
}

__device__ __forceinline__ float grid_step_dev(uint32_t biased_exp) { return __uint_as_float(biased_exp << 23); }

// quantises the padded boxes of the nk children of one node (same rules as the host's quantise_node)
__device__ void quantise_node_dev(BvhNode& n, int nk, const Box3* kb, float pad) {
    float nlo[3];
    n.scale_x = n.scale_y = n.scale_z = 1.0f;
    for (int a = 0; a < 3; ++a) {
        float lo = kb[0].lo[a] - pad, hi = kb[0].hi[a] + pad;
        for (int k = 1; k < nk; ++k) lo = fminf(lo, kb[k].lo[a] - pad), hi = fmaxf(hi, kb[k].hi[a] + pad);
        nlo[a] = lo;
        n.origin[a] = lo;
        const double ext = (double)hi - (double)lo;
        int e = ext > 0 ? (int)ceil(log2(ext / 255.0)) : -100;
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        for (;; ++e) {
            const float s = grid_step_dev((uint32_t)(e + 127));
            uint32_t lo_b = 0, hi_b = 0;
            bool ok = true;
            for (int k = 0; k < 4 && ok; ++k) {
                uint32_t ql = 255, qh = 0;
                if (k < nk) {
                    const float klo = kb[k].lo[a] - pad, khi = kb[k].hi[a] + pad;
                    const double fl = floor(((double)klo - (double)nlo[a]) / (double)s);
                    const double fh = ceil(((double)khi - (double)nlo[a]) / (double)s);
                    long il = (long)fmax(0.0, fmin(255.0, fl)), ih = (long)fmax(0.0, fmin(256.0, fh));
                    while (il > 0 && n.origin[a] + (float)il * s > klo) --il;
                    while (ih <= 255 && n.origin[a] + (float)ih * s < khi) ++ih;
                    if (ih > 255 || n.origin[a] + (float)il * s > klo) { ok = false; break; }
                    ql = (uint32_t)il, qh = (uint32_t)ih;
                }
                lo_b |= ql << (8 * k), hi_b |= qh << (8 * k);
            }
            if (ok) {
                n.q[2 * a] = lo_b, n.q[2 * a + 1] = hi_b;
                if (a == 0) n.scale_x = s;
                else if (a == 1) n.scale_y = s;
                else n.scale_z = s;
                break;
            }
            if (e > 120) break;
        }
    }
}

// frontier entry: binary internal node -> BVH4 node slot
__global__ void __launch_bounds__(128) k_emit(uint32_t n_in, const int2* __restrict__ frontier_in, int2* __restrict__ frontier_out,
                                               unsigned int* __restrict__ node_counter, unsigned int* __restrict__ out_count,
                                               const int* __restrict__ left, const int* __restrict__ right,
                                               const uint32_t* __restrict__ sorted_tri, const Box3* __restrict__ tri_box,
                                               const Box3* __restrict__ node_box, float pad, BvhNode* __restrict__ nodes) {
    const uint32_t i = blockIdx.x * 128u + threadIdx.x;
    if (i >= n_in) return;
    const int b = frontier_in[i].x, slot = frontier_in[i].y;
    int kids[4];
    int nk = 0;
    const int c2[2] = {left[b], right[b]};
    for (int k = 0; k < 2; ++k) {
        if (c2[k] < 0) kids[nk++] = c2[k];             // a triangle directly below
        else kids[nk++] = left[c2[k]], kids[nk++] = right[c2[k]]; // open the inner child: its two children
    }
    Box3 kb[4];
    for (int k = 0; k < nk; ++k) kb[k] = kids[k] >= 0 ? node_box[kids[k]] : tri_box[sorted_tri[~kids[k]]];
    BvhNode out;
    quantise_node_dev(out, nk, kb, pad);
    for (int k = 0; k < 4; ++k) {
        if (k >= nk) { out.child[k] = kChildEmpty; continue; }
        if (kids[k] < 0) {
            out.child[k] = ~(int32_t)(((uint32_t)~kids[k] << 2) | 0u); // leaf: one triangle at its Morton position
        } else {
            const unsigned int idx = atomicAdd(node_counter, 1u);
            out.child[k] = (int32_t)idx;
            frontier_out[atomicAdd(out_count, 1u)] = make_int2(kids[k], (int)idx);
        }
    }
    nodes[slot] = out;
}

__global__ void __launch_bounds__(256) k_permute(uint32_t n, const uint32_t* __restrict__ sorted_tri, const TriRec* __restrict__ gtris,
                                                  TriRec* __restrict__ tris) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p < n) tris[p] = gtris[sorted_tri[p]];
}

struct Tmp { // frees everything on every return path
    std::vector<void*> ptrs;
    ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    hipError_t alloc(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p), *out = (T*)p;
        return e;
    }
};

} // namespace

// Builds nodes/tris for the flattened triangles of `hs` (hs.wverts, gtris in global order) on the current
// device and downloads them into hs.nodes / hs.tris. Needs at least 8 triangles (tiny scenes use the host).
int build_lbvh_gpu(HostScene& hs, const std::vector<TriRec>& gtris, std::string& err) {
    const uint32_t n = (uint32_t)gtris.size();
    if (n < 8) { err = "internal: GPU LBVH needs at least 8 triangles"; return RT_ERR_INVALID; }
    Tmp tmp;
    float* d_wverts; Box3 *d_tri_box, *d_node_box; unsigned long long *d_keys, *d_keys_sorted; uint32_t *d_vals, *d_sorted;
    int *d_left, *d_right, *d_parent_inner, *d_parent_leaf; unsigned int *d_flags, *d_counters; int2 *d_front[2];
    TriRec *d_gtris, *d_tris; BvhNode* d_nodes;
    LB_CHK(tmp.alloc(&d_wverts, 9 * (size_t)n));
    LB_CHK(tmp.alloc(&d_tri_box, n)); LB_CHK(tmp.alloc(&d_node_box, n));
    LB_CHK(tmp.alloc(&d_keys, n)); LB_CHK(tmp.alloc(&d_keys_sorted, n)); LB_CHK(tmp.alloc(&d_vals, n)); LB_CHK(tmp.alloc(&d_sorted, n));
    LB_CHK(tmp.alloc(&d_left, n)); LB_CHK(tmp.alloc(&d_right, n)); LB_CHK(tmp.alloc(&d_parent_inner, n)); LB_CHK(tmp.alloc(&d_parent_leaf, n));
    LB_CHK(tmp.alloc(&d_flags, n)); LB_CHK(tmp.alloc(&d_counters, 4)); LB_CHK(tmp.alloc(&d_front[0], n)); LB_CHK(tmp.alloc(&d_front[1], n));
    LB_CHK(tmp.alloc(&d_gtris, n)); LB_CHK(tmp.alloc(&d_tris, n)); LB_CHK(tmp.alloc(&d_nodes, n));
    LB_CHK(hipMemcpy(d_wverts, hs.wverts.data(), 9 * (size_t)n * 4, hipMemcpyHostToDevice));
    LB_CHK(hipMemcpy(d_gtris, gtris.data(), (size_t)n * sizeof(TriRec), hipMemcpyHostToDevice));
    LB_CHK(hipMemset(d_flags, 0, (size_t)n * 4));

    // scene bounds (hs.bounds_*) are set by the caller; Morton grid over the bounds of the box centres ~ scene bounds
    float3 clo = make_float3(hs.bounds_lo[0], hs.bounds_lo[1], hs.bounds_lo[2]);
    float3 inv = make_float3(hs.bounds_hi[0] > hs.bounds_lo[0] ? 1.0f / (hs.bounds_hi[0] - hs.bounds_lo[0]) : 0.0f,
                             hs.bounds_hi[1] > hs.bounds_lo[1] ? 1.0f / (hs.bounds_hi[1] - hs.bounds_lo[1]) : 0.0f,
                             hs.bounds_hi[2] > hs.bounds_lo[2] ? 1.0f / (hs.bounds_hi[2] - hs.bounds_lo[2]) : 0.0f);
    const uint32_t g256 = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_prims, dim3(g256), dim3(256), 0, 0, n, d_wverts, clo, inv, d_tri_box, d_keys, d_vals);
    size_t sort_bytes = 0;
    LB_CHK(rocprim::radix_sort_pairs(nullptr, sort_bytes, d_keys, d_keys_sorted, d_vals, d_sorted, n, 0, 63, 0));
    void* d_sort_tmp = nullptr;
    LB_CHK(tmp.alloc((unsigned char**)&d_sort_tmp, sort_bytes));
    LB_CHK(rocprim::radix_sort_pairs(d_sort_tmp, sort_bytes, d_keys, d_keys_sorted, d_vals, d_sorted, n, 0, 63, 0));
    hipLaunchKernelGGL(k_karras, dim3(g256), dim3(256), 0, 0, (int)n, d_keys_sorted, d_left, d_right, d_parent_inner, d_parent_leaf);
    hipLaunchKernelGGL(k_refit, dim3(g256), dim3(256), 0, 0, (int)n, d_sorted, d_tri_box, d_left, d_right, d_parent_inner, d_parent_leaf,
                       d_node_box, d_flags);
    hipLaunchKernelGGL(k_permute, dim3(g256), dim3(256), 0, 0, n, d_sorted, d_gtris, d_tris);
    LB_CHK(hipGetLastError());

    // breadth-first emission, one launch per BVH4 level; counters: [0] nodes allocated, [1] next frontier length
    unsigned int h_counters[2] = {1u, 0u};
    int2 root = make_int2(0, 0);
    LB_CHK(hipMemcpy(d_front[0], &root, sizeof(root), hipMemcpyHostToDevice));
    uint32_t n_front = 1, level = 0;
    int cur = 0;
    while (n_front) {
        if (++level > 64) { err = "GPU LBVH deeper than 64 levels"; return RT_ERR_UNSUPPORTED; } // the caller falls back to a balanced host build
        h_counters[1] = 0;
        LB_CHK(hipMemcpy(d_counters, h_counters, sizeof(h_counters), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_emit, dim3((n_front + 127u) / 128u), dim3(128), 0, 0, n_front, d_front[cur], d_front[cur ^ 1], d_counters,
                           d_counters + 1, d_left, d_right, d_sorted, d_tri_box, d_node_box, hs.pad, d_nodes);
        LB_CHK(hipGetLastError());
        LB_CHK(hipMemcpy(h_counters, d_counters, sizeof(h_counters), hipMemcpyDeviceToHost));
        n_front = h_counters[1];
        cur ^= 1;
    }
    const uint32_t n_nodes = h_counters[0];
    hs.nodes.resize(n_nodes);
    hs.tris.resize(n);
    LB_CHK(hipMemcpy(hs.nodes.data(), d_nodes, (size_t)n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
    LB_CHK(hipMemcpy(hs.tris.data(), d_tris, (size_t)n * sizeof(TriRec), hipMemcpyDeviceToHost));
    hs.max_depth = level;
    hs.max_leaf_tris = 1;
    return RT_OK;
}

} // namespace rt
