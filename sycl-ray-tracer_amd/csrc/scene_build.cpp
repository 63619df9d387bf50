// scene_build.cpp — host side of rt_scene_create: validation, instance flattening, BVH build.
//
// Replaces what Embree does behind rtcCommitScene for the reference (src/scene.cpp:101-107,
// 406-439,487-507): a two-level instanced triangle BVH becomes ONE flattened world-space BVH2
// over all instance triangles. The image does not depend on the tree (closest t, ties to the
// lowest global triangle index), so the builder is free to optimise for traversal speed.
//
// Compiled with -ffp-contract=off: the world-space vertices and edges computed here are part of
// the arithmetic contract shared with the CPU oracle (instance transform row order, e1 = v1 - v0).
#include "scene_build.h"
#include "rt_knobs.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <deque>
#include <functional>
#include <queue>

namespace rt {
namespace {

struct GNode { // generic build node
    float lo[3], hi[3];
    int32_t left = -1, right = -1; // inner: both >= 0
    uint32_t first = 0, count = 0; // range in `order`
    uint32_t depth = 0;
};

// The builders work on REFERENCES: (triangle, box). Without pre-splitting there is one reference per triangle and its box is the
// triangle's; presplit() below replaces large triangles by several references with the boxes of the triangle's clipped pieces.
struct Builder {
    const std::vector<float>& wv; // 9 per triangle
    std::vector<uint32_t> order;   // reference ids, partitioned in place by the builders
    std::vector<GNode> nodes;
    std::vector<float> cent; // 3 per reference
    std::vector<float> tlo, thi; // per-reference boxes
    std::vector<uint32_t> ref_tri; // reference -> triangle
    uint32_t n_split = 0;          // triangles that were split

    explicit Builder(const std::vector<float>& w) : wv(w) {
        const size_t n = w.size() / 9;
        order.resize(n);
        cent.resize(3 * n);
        tlo.resize(3 * n);
        thi.resize(3 * n);
        ref_tri.resize(n);
        for (size_t t = 0; t < n; ++t) {
            order[t] = (uint32_t)t;
            ref_tri[t] = (uint32_t)t;
            for (int a = 0; a < 3; ++a) {
                float p0 = w[9 * t + a], p1 = w[9 * t + 3 + a], p2 = w[9 * t + 6 + a];
                tlo[3 * t + a] = std::min(p0, std::min(p1, p2));
                thi[3 * t + a] = std::max(p0, std::max(p1, p2));
                cent[3 * t + a] = 0.5f * (tlo[3 * t + a] + thi[3 * t + a]);
            }
        }
    }

    void range_bounds(uint32_t first, uint32_t count, float* lo, float* hi, float* clo, float* chi) const {
        for (int a = 0; a < 3; ++a) {
            lo[a] = clo[a] = std::numeric_limits<float>::infinity();
            hi[a] = chi[a] = -std::numeric_limits<float>::infinity();
        }
        for (uint32_t k = 0; k < count; ++k) {
            uint32_t t = order[first + k];
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], tlo[3 * t + a]);
                hi[a] = std::max(hi[a], thi[3 * t + a]);
                clo[a] = std::min(clo[a], cent[3 * t + a]);
                chi[a] = std::max(chi[a], cent[3 * t + a]);
            }
        }
    }
};

// ---- spatial pre-splitting of large triangles (early split clipping with a budget) ------------------------------------
// A triangle that is large against the nodes it ends up in drags their boxes over empty space (a wall made of two triangles, Sponza's
// floor): every ray through that space pays a node visit and a triangle test for nothing. Its remedy in a top-down builder is the spatial
// split (Stich et al. 2009); the form built here is the pre-pass of Ernst & Greiner 2007 with a budget, as production builders use it:
// the references with the FATTEST boxes (most enclosed volume) are cut at the spatial median of their longest axis — the triangle is clipped against the plane
// (Sutherland-Hodgman, in double), each piece gets the tight box of its polygon — until the budget of extra references is spent or
// nothing is left above the floor size. The tree is then built over the pieces; a leaf stores the TRIANGLE (each once per leaf).
// A triangle in several leaves cannot change a result: the closest hit is min t with ties to the lowest triangle index (rule R5),
// and testing the same triangle twice yields the same t. What it costs: more leaf records; what it buys: tighter boxes.
struct SplitRef {
    uint32_t tri;
    int n;            // polygon vertices
    double p[10][3];  // a triangle clipped by k axis-aligned planes has at most 3 + k vertices; pieces are cut at most ~7 times
    float lo[3], hi[3];
    float key;        // fatness of the box (below): what the heap orders by
};
struct SplitRefLess {
    bool operator()(const SplitRef* a, const SplitRef* b) const { return a->key < b->key; }
};

// What makes a reference worth splitting is the EMPTY VOLUME its box encloses, not its size: a wall of two axis-aligned triangles has a
// flat box that drags nothing along (and is cheaper to trace than a tessellated one), a large triangle that runs diagonally through space
// has a fat one. Key: the geometric mean of the box's three extents (0 for a flat box).
inline float fatness(const float* lo, const float* hi) {
    const float dx = std::max(hi[0] - lo[0], 0.0f), dy = std::max(hi[1] - lo[1], 0.0f), dz = std::max(hi[2] - lo[2], 0.0f);
    return std::cbrt(dx * dy * dz);
}

inline void poly_box(SplitRef& r, const float* plo, const float* phi) { // tight box of the polygon, rounded outward, inside the parent's
    for (int a = 0; a < 3; ++a) {
        double lo = r.p[0][a], hi = r.p[0][a];
        for (int i = 1; i < r.n; ++i) lo = std::min(lo, r.p[i][a]), hi = std::max(hi, r.p[i][a]);
        float flo = (float)lo, fhi = (float)hi;
        if ((double)flo > lo) flo = std::nextafter(flo, -INFINITY);
        if ((double)fhi < hi) fhi = std::nextafter(fhi, INFINITY);
        r.lo[a] = std::max(flo, plo[a]), r.hi[a] = std::min(fhi, phi[a]);
    }
    r.key = fatness(r.lo, r.hi);
}

// polygon `in` clipped to the half space (keep_below ? x[axis] <= pos : x[axis] >= pos)
inline int clip_poly(const double (*in)[3], int n, int axis, double pos, bool keep_below, double (*out)[3]) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const double* a = in[i];
        const double* b = in[(i + 1) % n];
        const double da = keep_below ? pos - a[axis] : a[axis] - pos, db = keep_below ? pos - b[axis] : b[axis] - pos;
        if (da >= 0.0) {
            if (m < 10) std::memcpy(out[m++], a, sizeof(double) * 3);
        }
        if ((da > 0.0 && db < 0.0) || (da < 0.0 && db > 0.0)) {
            const double t = da / (da - db);
            if (m < 10) {
                for (int k = 0; k < 3; ++k) out[m][k] = a[k] + t * (b[k] - a[k]);
                out[m][axis] = pos;
                ++m;
            }
        }
    }
    return m;
}

// budget: extra references as a fraction of the triangle count; floor_extent: boxes whose fatness is below this are left alone
void presplit(Builder& b, double budget, float floor_extent) {
    const size_t T = b.wv.size() / 9;
    const size_t max_refs = T + (size_t)((double)T * budget);
    if (T == 0 || max_refs == T) return;
    std::deque<SplitRef> pool; // stable addresses
    std::priority_queue<SplitRef*, std::vector<SplitRef*>, SplitRefLess> heap;
    std::vector<uint8_t> was_split(T, 0);
    for (size_t t = 0; t < T; ++t) {
        const float ext = fatness(&b.tlo[3 * t], &b.thi[3 * t]);
        if (!(ext > floor_extent)) continue;
        pool.emplace_back();
        SplitRef& r = pool.back();
        r.tri = (uint32_t)t, r.n = 3;
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) r.p[v][a] = (double)b.wv[9 * t + 3 * v + a];
        std::memcpy(r.lo, &b.tlo[3 * t], 12), std::memcpy(r.hi, &b.thi[3 * t], 12);
        r.key = ext;
        heap.push(&r);
    }
    if (heap.empty()) return;
    size_t n_refs = T;
    std::vector<const SplitRef*> done; // pieces that are final
    while (!heap.empty()) {
        SplitRef* r = heap.top();
        heap.pop();
        if (n_refs >= max_refs || !(r->key > floor_extent) || r->n >= 9) {
            done.push_back(r);
            continue;
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (r->hi[a] - r->lo[a] > r->hi[axis] - r->lo[axis]) axis = a;
        const double pos = 0.5 * ((double)r->lo[axis] + (double)r->hi[axis]);
        pool.emplace_back();
        SplitRef& l = pool.back();
        pool.emplace_back();
        SplitRef& h = pool.back();
        l.tri = h.tri = r->tri;
        l.n = clip_poly(r->p, r->n, axis, pos, true, l.p);
        h.n = clip_poly(r->p, r->n, axis, pos, false, h.p);
        if (l.n < 3 || h.n < 3) { // the plane does not cut the polygon (degenerate): leave the piece as it is
            r->n = 9;             // (never asked again)
            done.push_back(r);
            continue;
        }
        float llo[3], lhi[3], hlo[3], hhi[3];
        std::memcpy(llo, r->lo, 12), std::memcpy(lhi, r->hi, 12), std::memcpy(hlo, r->lo, 12), std::memcpy(hhi, r->hi, 12);
        lhi[axis] = std::nextafter((float)pos, INFINITY), hlo[axis] = std::nextafter((float)pos, -INFINITY);
        poly_box(l, llo, lhi), poly_box(h, hlo, hhi);
        was_split[r->tri] = 1;
        ++n_refs;
        heap.push(&l), heap.push(&h);
    }
    // the reference arrays: unsplit triangles keep their slot; the pieces of split triangles are appended (the first piece takes the slot)
    std::vector<uint8_t> slot_used(T, 0);
    for (const SplitRef* r : done) {
        if (!was_split[r->tri]) continue; // popped but never cut: the triangle's own reference stands
        size_t id;
        if (!slot_used[r->tri]) id = r->tri, slot_used[r->tri] = 1;
        else {
            id = b.ref_tri.size();
            b.ref_tri.push_back(r->tri);
            b.order.push_back((uint32_t)id);
            b.tlo.resize(3 * (id + 1)), b.thi.resize(3 * (id + 1)), b.cent.resize(3 * (id + 1));
        }
        for (int a = 0; a < 3; ++a) b.tlo[3 * id + a] = r->lo[a], b.thi[3 * id + a] = r->hi[a], b.cent[3 * id + a] = 0.5f * (r->lo[a] + r->hi[a]);
    }
    for (size_t t = 0; t < T; ++t) b.n_split += was_split[t];
}

inline float half_area(const float* lo, const float* hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}

// ---- LBVH: 63-bit Morton order + top-down split at the highest differing bit -----------------------
inline uint64_t expand21(uint64_t v) { // spread the low 21 bits to every third bit
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

void build_lbvh(Builder& b) {
    const uint32_t n = (uint32_t)b.order.size();
    float lo[3], hi[3], clo[3], chi[3];
    b.range_bounds(0, n, lo, hi, clo, chi);
    std::vector<std::pair<uint64_t, uint32_t>> keys(n);
    for (uint32_t t = 0; t < n; ++t) {
        uint64_t code = 0;
        for (int a = 0; a < 3; ++a) {
            double ext = (double)chi[a] - (double)clo[a];
            double f = ext > 0 ? ((double)b.cent[3 * t + a] - (double)clo[a]) / ext : 0.0;
            uint64_t q = (uint64_t)std::min(2097151.0, std::max(0.0, f * 2097152.0));
            code |= expand21(q) << (2 - a);
        }
        keys[t] = {code, t};
    }
    std::sort(keys.begin(), keys.end());
    for (uint32_t i = 0; i < n; ++i) b.order[i] = keys[i].second;

    struct Work {
        int32_t node;
        uint32_t first, count, depth;
    };
    std::vector<Work> todo;
    b.nodes.clear();
    b.nodes.push_back(GNode{});
    todo.push_back({0, 0, n, 0});
    while (!todo.empty()) {
        Work w = todo.back();
        todo.pop_back();
        GNode nd;
        float c0[3], c1[3];
        b.range_bounds(w.first, w.count, nd.lo, nd.hi, c0, c1);
        nd.first = w.first, nd.count = w.count, nd.depth = w.depth;
        if (w.count > (uint32_t)kMaxLeafTris) {
            uint64_t ka = keys[w.first].first, kb = keys[w.first + w.count - 1].first;
            uint32_t split;
            if (ka == kb || w.depth >= 56) {
                split = w.count / 2; // identical codes (or runaway depth): halve the index range
            } else {
                int bit = 63 - __builtin_clzll(ka ^ kb);
                uint64_t mask = 1ull << bit;
                // first element whose `bit` is set (codes share all higher bits, so it is monotone)
                uint32_t l = 0, r = w.count;
                while (l < r) {
                    uint32_t m = (l + r) / 2;
                    if (keys[w.first + m].first & mask) r = m; else l = m + 1;
                }
                split = l;
                if (split == 0 || split == w.count) split = w.count / 2;
            }
            nd.left = (int32_t)b.nodes.size();
            nd.right = nd.left + 1;
            b.nodes.push_back(GNode{});
            b.nodes.push_back(GNode{});
            todo.push_back({nd.left, w.first, split, w.depth + 1});
            todo.push_back({nd.right, w.first + split, w.count - split, w.depth + 1});
        }
        b.nodes[w.node] = nd;
    }
}

// ---- binned SAH, top-down -------------------------------------------------------------------------
void build_sah(Builder& b, uint32_t max_sah_depth) {
    const uint32_t n = (uint32_t)b.order.size();
    constexpr int NB = 32;
    // cost of one traversal step relative to one triangle test, for the leaf-vs-split decision of small ranges
    float trav_cost = 1.0f;
    if (const char* e = dev_knob("RT_SAH_TRAV_COST")) trav_cost = (float)std::atof(e);
    struct Work {
        int32_t node;
        uint32_t first, count, depth;
    };
    std::vector<Work> todo;
    b.nodes.clear();
    b.nodes.push_back(GNode{});
    todo.push_back({0, 0, n, 0});
    std::vector<uint32_t> tmp;
    while (!todo.empty()) {
        Work w = todo.back();
        todo.pop_back();
        GNode nd;
        float clo[3], chi[3];
        b.range_bounds(w.first, w.count, nd.lo, nd.hi, clo, chi);
        nd.first = w.first, nd.count = w.count, nd.depth = w.depth;
        bool make_leaf = w.count == 1;
        uint32_t split = 0;
        if (!make_leaf) {
            const float parent_area = half_area(nd.lo, nd.hi);
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            if (w.depth < max_sah_depth) {
                for (int a = 0; a < 3; ++a) {
                    float ext = chi[a] - clo[a];
                    if (!(ext > 0.0f)) continue;
                    float scale = (float)NB / ext;
                    uint32_t cnt[NB] = {0};
                    float blo[NB][3], bhi[NB][3];
                    for (int i = 0; i < NB; ++i)
                        for (int k = 0; k < 3; ++k) blo[i][k] = INFINITY, bhi[i][k] = -INFINITY;
                    for (uint32_t k = 0; k < w.count; ++k) {
                        uint32_t t = b.order[w.first + k];
                        int bi = std::min(NB - 1, std::max(0, (int)((b.cent[3 * t + a] - clo[a]) * scale)));
                        cnt[bi]++;
                        for (int j = 0; j < 3; ++j) {
                            blo[bi][j] = std::min(blo[bi][j], b.tlo[3 * t + j]);
                            bhi[bi][j] = std::max(bhi[bi][j], b.thi[3 * t + j]);
                        }
                    }
                    float ra[NB];
                    uint32_t rc[NB];
                    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                    uint32_t c = 0;
                    for (int i = NB - 1; i > 0; --i) {
                        c += cnt[i];
                        for (int j = 0; j < 3; ++j) lo[j] = std::min(lo[j], blo[i][j]), hi[j] = std::max(hi[j], bhi[i][j]);
                        ra[i] = half_area(lo, hi), rc[i] = c;
                    }
                    for (int j = 0; j < 3; ++j) lo[j] = INFINITY, hi[j] = -INFINITY;
                    c = 0;
                    for (int i = 0; i < NB - 1; ++i) {
                        c += cnt[i];
                        for (int j = 0; j < 3; ++j) lo[j] = std::min(lo[j], blo[i][j]), hi[j] = std::max(hi[j], bhi[i][j]);
                        if (c == 0 || rc[i + 1] == 0) continue;
                        float cost = half_area(lo, hi) * (float)c + ra[i + 1] * (float)rc[i + 1];
                        if (cost < best_cost) best_cost = cost, best_axis = a, best_bin = i;
                    }
                }
            }
            // leaf cost = count * area; split cost = 1 traversal step * area + children
            if (w.count <= (uint32_t)kMaxLeafTris &&
                !(best_axis >= 0 && best_cost + trav_cost * parent_area < (float)w.count * parent_area)) {
                make_leaf = true;
            } else if (best_axis >= 0) {
                float ext = chi[best_axis] - clo[best_axis];
                float scale = (float)NB / ext;
                auto mid = std::partition(b.order.begin() + w.first, b.order.begin() + w.first + w.count, [&](uint32_t t) {
                    int bi = std::min(NB - 1, std::max(0, (int)((b.cent[3 * t + best_axis] - clo[best_axis]) * scale)));
                    return bi <= best_bin;
                });
                split = (uint32_t)(mid - (b.order.begin() + w.first));
            }
            if (!make_leaf && (split == 0 || split == w.count)) {
                // all centroids coincide (or depth guard): median split on the index order
                split = w.count / 2;
                std::nth_element(b.order.begin() + w.first, b.order.begin() + w.first + split,
                                 b.order.begin() + w.first + w.count);
            }
        }
        if (!make_leaf) {
            nd.left = (int32_t)b.nodes.size();
            nd.right = nd.left + 1;
            b.nodes.push_back(GNode{});
            b.nodes.push_back(GNode{});
            todo.push_back({nd.left, w.first, split, w.depth + 1});
            todo.push_back({nd.right, w.first + split, w.count - split, w.depth + 1});
        }
        b.nodes[w.node] = nd;
    }
}

// ---- insertion-based optimisation of the binary tree (after Bittner, Hapala, Havran 2013) ---------------------
// A top-down SAH build decides every split greedily and never revisits it. This pass takes inner nodes in order of
// decreasing surface area, unlinks the node (its two subtrees become orphans, its parent is replaced by its
// sibling) and re-inserts both subtrees where they increase the tree's total surface area least, found by a
// branch-and-bound search from the root. Leaves (triangle ranges) are untouched: only the topology above them
// changes, so the traversal result cannot change, only the number of nodes a ray visits.
void optimise_by_reinsertion(Builder& b, int passes, double fraction) {
    std::vector<GNode>& n = b.nodes;
    const int32_t count = (int32_t)n.size();
    if (count < 7) return;
    std::vector<int32_t> parent((size_t)count, -1);
    for (int32_t i = 0; i < count; ++i)
        if (n[i].left >= 0) parent[n[i].left] = i, parent[n[i].right] = i;
    auto area = [&](int32_t i) { return half_area(n[i].lo, n[i].hi); };
    auto merge = [&](int32_t dst, int32_t a, int32_t c) {
        for (int k = 0; k < 3; ++k) n[dst].lo[k] = std::min(n[a].lo[k], n[c].lo[k]), n[dst].hi[k] = std::max(n[a].hi[k], n[c].hi[k]);
    };
    auto refit_up = [&](int32_t i) {
        for (; i >= 0; i = parent[i]) {
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k)
                lo[k] = std::min(n[n[i].left].lo[k], n[n[i].right].lo[k]), hi[k] = std::max(n[n[i].left].hi[k], n[n[i].right].hi[k]);
            if (std::memcmp(lo, n[i].lo, 12) == 0 && std::memcmp(hi, n[i].hi, 12) == 0) break;
            std::memcpy(n[i].lo, lo, 12), std::memcpy(n[i].hi, hi, 12);
        }
    };
    auto union_area = [&](int32_t a, int32_t c) {
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) lo[k] = std::min(n[a].lo[k], n[c].lo[k]), hi[k] = std::max(n[a].hi[k], n[c].hi[k]);
        return half_area(lo, hi);
    };
    struct Cand { float bound; int32_t node; float induced; };
    auto cmp = [](const Cand& x, const Cand& y) { return x.bound > y.bound; };
    std::vector<Cand> heap;
    // the node below which `x` costs least: cost = area(node ∪ x) + the growth of every ancestor of the new parent
    auto find_best = [&](int32_t x) {
        const float ax = area(x);
        float best = std::numeric_limits<float>::infinity();
        int32_t best_node = -1;
        heap.clear();
        for (int32_t c : {n[0].left, n[0].right}) heap.push_back({0.0f, c, 0.0f}); // never AT the root: node 0 stays the root
        std::make_heap(heap.begin(), heap.end(), cmp);
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            const Cand c = heap.back();
            heap.pop_back();
            if (c.bound >= best) break;
            const float direct = union_area(c.node, x);
            const float total = c.induced + direct;
            if (total < best) best = total, best_node = c.node;
            const float child_induced = total - area(c.node);
            if (n[c.node].left >= 0 && child_induced + ax < best) {
                heap.push_back({child_induced + ax, n[c.node].left, child_induced});
                std::push_heap(heap.begin(), heap.end(), cmp);
                heap.push_back({child_induced + ax, n[c.node].right, child_induced});
                std::push_heap(heap.begin(), heap.end(), cmp);
            }
        }
        return best_node;
    };
    std::vector<int32_t> order;
    for (int pass = 0; pass < passes; ++pass) {
        order.clear();
        for (int32_t i = 1; i < count; ++i)
            if (n[i].left >= 0 && parent[i] > 0) order.push_back(i); // inner, not the root, not a child of the root
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t c) { return area(a) > area(c); });
        order.resize((size_t)((double)order.size() * fraction));
        for (int32_t N : order) {
            const int32_t P = parent[N];
            if (n[N].left < 0 || P <= 0) continue; // the tree changed under us: N became a root child or was recycled
            const int32_t G = parent[P];
            const int32_t S = n[P].left == N ? n[P].right : n[P].left;
            int32_t X[2] = {n[N].left, n[N].right};
            if (area(X[0]) < area(X[1])) std::swap(X[0], X[1]);
            // unlink N and P: G adopts S
            (n[G].left == P ? n[G].left : n[G].right) = S;
            parent[S] = G;
            refit_up(G);
            int32_t spare[2] = {P, N};
            for (int k = 0; k < 2; ++k) {
                const int32_t x = X[k], F = spare[k];
                const int32_t B = find_best(x);
                const int32_t PB = parent[B];
                (n[PB].left == B ? n[PB].left : n[PB].right) = F;
                parent[F] = PB;
                n[F].left = B, n[F].right = x, n[F].first = 0, n[F].count = 0;
                parent[B] = F, parent[x] = F;
                merge(F, B, x);
                refit_up(PB);
            }
        }
    }
    // depths (statistics only) from the new topology
    std::vector<int32_t> stack{0};
    n[0].depth = 0;
    while (!stack.empty()) {
        const int32_t i = stack.back();
        stack.pop_back();
        if (n[i].left >= 0) {
            n[n[i].left].depth = n[n[i].right].depth = n[i].depth + 1;
            stack.push_back(n[i].left), stack.push_back(n[i].right);
        }
    }
}

inline float grid_step(uint32_t biased_exp) { // 2^(e-127) as a float, exactly as the kernel decodes it
    uint32_t bits = biased_exp << 23;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// Quantises the (already padded) child boxes of one BVH4 node. Returns false only on non-finite input.
bool quantise_node(BvhNode& n, int nk, const float (*klo)[3], const float (*khi)[3]) {
    float nlo[3], nhi[3];
    for (int a = 0; a < 3; ++a) {
        nlo[a] = klo[0][a], nhi[a] = khi[0][a];
        for (int k = 1; k < nk; ++k) nlo[a] = std::min(nlo[a], klo[k][a]), nhi[a] = std::max(nhi[a], khi[k][a]);
        if (!std::isfinite(nlo[a]) || !std::isfinite(nhi[a])) return false;
        n.origin[a] = nlo[a];
    }
    uint32_t q[6] = {0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 3; ++a) {
        const double ext = (double)nhi[a] - (double)nlo[a];
        int e = ext > 0 ? (int)std::ceil(std::log2(ext / 255.0)) : -100;
        e = std::max(-100, std::min(100, e));
        for (;; ++e) { // raise the grid step until every plane fits in 8 bits
            const float s = grid_step((uint32_t)(e + 127));
            uint32_t lo_b = 0, hi_b = 0;
            bool ok = true;
            for (int k = 0; k < 4 && ok; ++k) {
                uint32_t ql = 255, qh = 0; // absent child: inverted box = a miss for every ray inside the contract's range (rt_device.h, trav_inner)
                if (k < nk) {
                    double fl = std::floor(((double)klo[k][a] - (double)nlo[a]) / (double)s);
                    double fh = std::ceil(((double)khi[k][a] - (double)nlo[a]) / (double)s);
                    long il = (long)std::max(0.0, std::min(255.0, fl)), ih = (long)std::max(0.0, std::min(256.0, fh));
                    // verify with the kernel's own float decode; nudge outwards if rounding bit us
                    while (il > 0 && n.origin[a] + (float)il * s > klo[k][a]) --il;
                    while (ih <= 255 && n.origin[a] + (float)ih * s < khi[k][a]) ++ih;
                    if (ih > 255 || n.origin[a] + (float)il * s > klo[k][a]) { ok = false; break; }
                    ql = (uint32_t)il, qh = (uint32_t)ih;
                }
                lo_b |= ql << (8 * k), hi_b |= qh << (8 * k);
            }
            if (ok) {
                q[2 * a] = lo_b, q[2 * a + 1] = hi_b;
                n.set_scale(a, s);
                break;
            }
            if (e > 120) return false;
        }
    }
    for (int i = 0; i < 6; ++i) n.q[i] = q[i];
    return true;
}

} // namespace

// Which binary nodes become BVH4 nodes: dynamic programme over the binary tree (after Ylitie, Karras, Laine 2017).
// best[n][k] = least SAH cost of covering subtree n with at most k child slots of its BVH4 parent; a subtree that
// takes one slot is a leaf or a BVH4 node of its own (area + the best way to hand its 4 slots to its two halves).
// cut[n][4] = slots a BVH4 node n gives to its LEFT half. best[0][1] / area(root) is the 4-wide tree's surface-area cost.
static void collapse_dp(const Builder& b, std::vector<std::array<float, 5>>& best, std::vector<std::array<uint8_t, 5>>& cut) {
    best.assign(b.nodes.size(), {}), cut.assign(b.nodes.size(), {});
    std::vector<int32_t> post, st{0};
    while (!st.empty()) { // pre-order, reversed = children before parents
        const int32_t i = st.back();
        st.pop_back();
        post.push_back(i);
        if (b.nodes[i].left >= 0) st.push_back(b.nodes[i].left), st.push_back(b.nodes[i].right);
    }
    for (size_t q = post.size(); q-- > 0;) {
        const int32_t i = post[q];
        const GNode& g = b.nodes[i];
        const float ar = half_area(g.lo, g.hi);
        if (g.left < 0) {
            for (int k = 1; k <= 4; ++k) best[i][k] = ar * (float)g.count;
            continue;
        }
        float dist[5];
        uint8_t dl[5];
        for (int k = 2; k <= 4; ++k) {
            dist[k] = std::numeric_limits<float>::infinity(), dl[k] = 1;
            for (int l = 1; l < k; ++l) {
                const float c = best[g.left][l] + best[g.right][k - l];
                if (c < dist[k]) dist[k] = c, dl[k] = (uint8_t)l;
            }
        }
        best[i][1] = ar + dist[4];
        for (int k = 2; k <= 4; ++k) best[i][k] = std::min(dist[k], best[i][k - 1]);
        cut[i][4] = dl[4]; // what the node does with its OWN four slots when it is a BVH4 node (best[i][1])
    }
}

int build_host_scene(const rt_scene_desc* d, int bvh_kind, HostScene& hs, std::string& err) {
    if (!d) { err = "null scene description"; return RT_ERR_INVALID; }
    if (d->n_triangles && (!d->positions || !d->normals || !d->uvs || !d->indices || !d->tri_instance)) {
        err = "null geometry array"; return RT_ERR_INVALID;
    }
    if ((d->n_instances && !d->instances) || (d->n_materials && !d->materials)) {
        err = "null instance/material array"; return RT_ERR_INVALID;
    }
    if (d->n_layers > RT_TEX_MAX_LAYERS) { err = "more than 128 texture layers"; return RT_ERR_INVALID; }
    if (d->n_layers && !d->textures) { err = "null texture array"; return RT_ERR_INVALID; }
    if (d->n_triangles >= (1u << 28)) { err = "too many triangles (limit 2^28)"; return RT_ERR_INVALID; }
    if (bvh_kind != RT_BVH_DEFAULT && bvh_kind != RT_BVH_LBVH && bvh_kind != RT_BVH_SAH && bvh_kind != RT_BVH_MEDIAN_INTERNAL &&
        bvh_kind != RT_BVH_LBVH_GPU) {
        err = "unknown bvh_kind"; return RT_ERR_INVALID;
    }
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        if (d->instances[i].material >= d->n_materials) { err = "instance material out of range"; return RT_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rt_material& m = d->materials[i];
        if (m.type > RT_MAT_DIELECTRIC || m.tex_kind > RT_TEX_IMAGE) { err = "bad material type"; return RT_ERR_INVALID; }
        if (m.tex_kind == RT_TEX_IMAGE && m.type != RT_MAT_DIELECTRIC && m.type != RT_MAT_NONE &&
            m.tex_layer >= d->n_layers) { err = "material texture layer out of range"; return RT_ERR_INVALID; }
    }
    const uint32_t T = d->n_triangles;
    for (uint32_t t = 0; t < T; ++t) {
        if (d->tri_instance[t] >= d->n_instances) { err = "tri_instance out of range"; return RT_ERR_INVALID; }
        for (int k = 0; k < 3; ++k)
            if (d->indices[3 * t + k] >= d->n_vertices) { err = "vertex index out of range"; return RT_ERR_INVALID; }
    }

    hs = HostScene{};
    hs.sky[0] = d->sky[0], hs.sky[1] = d->sky[1], hs.sky[2] = d->sky[2];
    hs.n_layers = d->n_layers;
    if (d->n_layers) hs.tex.assign(d->textures, d->textures + (size_t)d->n_layers * 512u * 512u * 4u);
    hs.packed_mat = d->n_instances <= (1u << kPackedInstBits) && d->n_materials <= (1u << (32u - kPackedInstBits));
    // The device's instance table. Packed shading word (the rule, rt_types.h): the material index travels in the word, so what is left of an
    // instance is its normal matrix — and a scene has far fewer DISTINCT matrices than instances (one instance per glTF node and primitive:
    // 166 and 5 on the atrium). The table holds every distinct bit pattern once, ordered by the number of triangles that use it, so that the
    // head of the table — what the shading kernels stage in LDS — covers the most hits. inst_slot[i]: instance i's row of the table.
    std::vector<uint32_t> inst_slot(d->n_instances);
    if (hs.packed_mat) {
        std::vector<uint64_t> use(d->n_instances, 0);
        for (uint32_t t = 0; t < d->n_triangles; ++t) use[d->tri_instance[t]]++;
        std::map<std::array<uint32_t, 9>, uint32_t> seen; // bit pattern -> provisional row
        std::vector<std::array<uint32_t, 9>> rows;
        std::vector<uint64_t> row_use;
        for (uint32_t i = 0; i < d->n_instances; ++i) {
            std::array<uint32_t, 9> key;
            std::memcpy(key.data(), d->instances[i].normal_mat, 36);
            auto it = seen.find(key);
            if (it == seen.end()) {
                it = seen.emplace(key, (uint32_t)rows.size()).first;
                rows.push_back(key), row_use.push_back(0);
            }
            inst_slot[i] = it->second;
            row_use[it->second] += use[i];
        }
        std::vector<uint32_t> order(rows.size()), rank(rows.size());
        for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return row_use[a] > row_use[b]; });
        for (uint32_t k = 0; k < order.size(); ++k) rank[order[k]] = k;
        hs.inst.assign(rows.size(), InstRec{});
        for (uint32_t k = 0; k < rows.size(); ++k) std::memcpy(hs.inst[rank[k]].normal_mat, rows[k].data(), 36);
        for (uint32_t i = 0; i < d->n_instances; ++i) inst_slot[i] = rank[inst_slot[i]];
    } else {
        hs.inst.resize(d->n_instances);
        for (uint32_t i = 0; i < d->n_instances; ++i) {
            std::memcpy(hs.inst[i].normal_mat, d->instances[i].normal_mat, sizeof(float) * 9);
            hs.inst[i].material = d->instances[i].material;
            hs.inst[i].pad[0] = hs.inst[i].pad[1] = 0;
            inst_slot[i] = i;
        }
    }
    hs.mats.resize(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rt_material& m = d->materials[i];
        MatRec& r = hs.mats[i];
        r.type = m.type, r.tex_kind = m.tex_kind, r.tex_layer = m.tex_layer;
        std::memcpy(r.color, m.color, 12), std::memcpy(r.emissive, m.emissive, 12);
        r.roughness = m.roughness, r.ior = m.ior, r.pad = 0;
    }

    // flatten: world = ((m0*x + m4*y) + m8*z) + m12 per row of the column-major instance matrix
    hs.wverts.resize(9 * (size_t)T);
    hs.shade.resize(T);
    std::vector<TriRec> gtris(T);
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t ii = d->tri_instance[t];
        const float* m = d->instances[ii].transform;
        float p[3][3];
        for (int k = 0; k < 3; ++k) {
            const uint32_t vi = d->indices[3 * t + k];
            const float x = d->positions[3 * vi], y = d->positions[3 * vi + 1], z = d->positions[3 * vi + 2];
            p[k][0] = ((m[0] * x + m[4] * y) + m[8] * z) + m[12];
            p[k][1] = ((m[1] * x + m[5] * y) + m[9] * z) + m[13];
            p[k][2] = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
            for (int a = 0; a < 3; ++a) hs.wverts[9 * (size_t)t + 3 * k + a] = p[k][a];
        }
        TriRec& tr = gtris[t];
        for (int a = 0; a < 3; ++a) {
            tr.v0[a] = p[0][a];
            tr.e1[a] = p[1][a] - p[0][a];
            tr.e2[a] = p[2][a] - p[0][a];
        }
        tr.global_index = t, tr.pad[0] = tr.pad[1] = 0;
        ShadeRec& s = hs.shade[t];
        const uint32_t i0 = d->indices[3 * t], i1 = d->indices[3 * t + 1], i2 = d->indices[3 * t + 2];
        std::memcpy(s.n0, d->normals + 3 * i0, 12), std::memcpy(s.n1, d->normals + 3 * i1, 12);
        std::memcpy(s.n2, d->normals + 3 * i2, 12);
        std::memcpy(s.uv0, d->uvs + 2 * i0, 8), std::memcpy(s.uv1, d->uvs + 2 * i1, 8);
        std::memcpy(s.uv2, d->uvs + 2 * i2, 8);
        s.instance = hs.packed_mat ? (inst_slot[ii] | (d->instances[ii].material << kPackedInstBits)) : ii;
    }
    for (size_t i = 0; i < hs.wverts.size(); ++i) {
        if (!std::isfinite(hs.wverts[i])) { err = "non-finite world-space vertex"; return RT_ERR_INVALID; }
    }

    auto empty_node = []() {
        BvhNode n{};
        n.scale_x = n.scale_y = n.scale_z = 1.0f;
        for (int i = 0; i < 6; ++i) n.q[i] = (i & 1) ? 0u : 0xFFFFFFFFu; // qlo = 255, qhi = 0
        for (int k = 0; k < 4; ++k) n.child[k] = kChildEmpty;
        return n;
    };
    if (T == 0) {
        hs.nodes.assign(1, empty_node());
        hs.tris.assign(1, TriRec{}); // one dummy record so device pointers are never null
        hs.tris[0].global_index = kNoTri;
        return RT_OK;
    }

    for (int a = 0; a < 3; ++a) hs.bounds_lo[a] = INFINITY, hs.bounds_hi[a] = -INFINITY;
    for (size_t i = 0; i < hs.wverts.size(); ++i) {
        const int a = (int)(i % 3);
        hs.bounds_lo[a] = std::min(hs.bounds_lo[a], hs.wverts[i]), hs.bounds_hi[a] = std::max(hs.bounds_hi[a], hs.wverts[i]);
    }
    float ext = 0.0f, amax = 0.0f;
    for (int a = 0; a < 3; ++a) {
        ext = std::max(ext, hs.bounds_hi[a] - hs.bounds_lo[a]);
        amax = std::max(amax, std::max(std::fabs(hs.bounds_lo[a]), std::fabs(hs.bounds_hi[a])));
    }
    // Absolute padding of every box: 2e-5 x scene scale. Float rounding of the slab test and of the
    // Moller-Trumbore t is ~1e-7 x scale, so a padded box can only cull triangles the exact test
    // would also miss ("culling is conservative").
    hs.pad = 2e-5f * std::max(ext, amax) + 1e-30f;

    // Exact worst-case traversal stack need: at every level all siblings of the entered child may be waiting.
    std::function<uint32_t(int32_t)> need = [&](int32_t node) -> uint32_t {
        const BvhNode& n = hs.nodes[node];
        uint32_t nk = 0, worst = 0;
        for (int k = 0; k < 4; ++k) nk += n.child[k] != kChildEmpty;
        for (int k = 0; k < 4; ++k)
            if (n.child[k] != kChildEmpty) worst = std::max(worst, (nk - 1) + (n.child[k] >= 0 ? need(n.child[k]) : 0u));
        return worst;
    };

    if (bvh_kind == RT_BVH_LBVH_GPU && T >= 8) { // whole build on the device (lbvh_gpu.hip); tiny scenes use the host
        int rc = build_lbvh_gpu(hs, gtris, err);
        if (rc == RT_ERR_UNSUPPORTED) { // degenerate Morton order (deeper than the emission loop allows): balanced host fallback
            err.clear();
            return build_host_scene(d, RT_BVH_MEDIAN_INTERNAL, hs, err);
        }
        if (rc != RT_OK) return rc;
        hs.stack_need = need(0);
        if (hs.stack_need + 2 >= (uint32_t)kStackSize) return build_host_scene(d, RT_BVH_MEDIAN_INTERNAL, hs, err);
        // surface-area cost of the downloaded tree (the diagnostic rt_scene_info reports for the host builders too): every node's
        // children boxes decoded, inner child 1 step x area, leaf child (triangles) x area, relative to the root's area
        {
            double cost = 0.0, root_area = 0.0;
            for (size_t i = 0; i < hs.nodes.size(); ++i) {
                const BvhNode& n = hs.nodes[i];
                float nlo[3] = {INFINITY, INFINITY, INFINITY}, nhi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int k = 0; k < 4; ++k) {
                    if (n.child[k] == kChildEmpty) continue;
                    float lo[3], hi[3];
                    for (int a = 0; a < 3; ++a) {
                        lo[a] = n.origin[a] + (float)((n.q[2 * a] >> (8 * k)) & 0xffu) * n.scale(a);
                        hi[a] = n.origin[a] + (float)((n.q[2 * a + 1] >> (8 * k)) & 0xffu) * n.scale(a);
                        nlo[a] = std::min(nlo[a], lo[a]), nhi[a] = std::max(nhi[a], hi[a]);
                    }
                    const double w = n.child[k] >= 0 ? 1.0 : (double)(((uint32_t)~n.child[k] & 3u) + 1u);
                    cost += w * (double)half_area(lo, hi);
                }
                if (i == 0) root_area = (double)half_area(nlo, nhi);
            }
            hs.sah_cost = root_area > 0.0 ? 1.0 + cost / root_area : 0.0;
        }
        return RT_OK;
    }

    std::unique_ptr<Builder> bp(new Builder(hs.wverts));
    if (bvh_kind == RT_BVH_LBVH || bvh_kind == RT_BVH_LBVH_GPU) build_lbvh(*bp);
    else {
        // pre-splitting of large triangles (SAH builder only): at most 30 % more references, only boxes FATTER (geometric mean of the three
        // extents) than 1/64 of the scene's largest extent — the finely tessellated bench scene has none and neither have axis-aligned walls
        // (their trees are unchanged); large triangles that cross space diagonally do. Where it did split, the tree without splits is
        // built too and the one with the lower surface-area cost is kept (a 116-triangle Cornell box gets worse, a tilted hall 15 % better).
        // RT_BVH_SPLIT_BUDGET=0 switches it off, RT_BVH_SPLIT_FLOOR=<fraction of the scene extent> moves the floor.
        double budget = 0.3, floor_frac = 1.0 / 64.0;
        if (const char* e = dev_knob("RT_BVH_SPLIT_BUDGET")) budget = std::max(0.0, std::min(4.0, std::atof(e)));
        if (const char* e = dev_knob("RT_BVH_SPLIT_FLOOR")) floor_frac = std::max(1e-4, std::min(1.0, std::atof(e)));
        // one pass over the larger half of the inner nodes: -2.6 % SAH cost, 17.9 -> 17.5 node visits per ray on the
        // atrium (+0.7 % rays/s) for 0.18 s of extra build time; further passes change nothing. RT_BVH_REINSERT=0: off
        int passes = 1;
        double fraction = 0.5;
        if (const char* e = dev_knob("RT_BVH_REINSERT")) passes = std::atoi(e);
        if (const char* e = dev_knob("RT_BVH_REINSERT_FRACTION")) fraction = std::atof(e);
        auto run = [&](Builder& bb, bool split) {
            if (split && bvh_kind != RT_BVH_MEDIAN_INTERNAL && budget > 0.0) presplit(bb, budget, (float)(floor_frac * (double)ext));
            build_sah(bb, bvh_kind == RT_BVH_MEDIAN_INTERNAL ? 0u : 48u);
            if (bvh_kind != RT_BVH_MEDIAN_INTERNAL && passes > 0) optimise_by_reinsertion(bb, passes, std::min(1.0, std::max(0.0, fraction)));
        };
        auto wide_cost = [](const Builder& bb) { // surface-area cost of the 4-wide tree the collapse below will make of this binary tree
            std::vector<std::array<float, 5>> bst;
            std::vector<std::array<uint8_t, 5>> ct;
            collapse_dp(bb, bst, ct);
            return (double)bst[0][1] / std::max(1e-30, (double)half_area(bb.nodes[0].lo, bb.nodes[0].hi));
        };
        run(*bp, true);
        if (bp->n_split > 0) {
            std::unique_ptr<Builder> plain(new Builder(hs.wverts));
            run(*plain, false);
            if (bp->nodes[0].left < 0 || plain->nodes[0].left < 0 || wide_cost(*plain) <= wide_cost(*bp)) bp = std::move(plain);
        }
    }
    Builder& b = *bp;
    const GNode& g0 = b.nodes[0];

    // Collapse the binary tree into a 4-wide one and emit it top first (the order is described below): the nodes the traversal kernels
    // stage in LDS are contiguous at the head of the array.
    hs.tris.clear();
    hs.tris.reserve(b.order.size());
    hs.nodes.clear();
    auto leaf_code = [&](const GNode& g, uint32_t depth) -> int32_t {
        uint32_t first = (uint32_t)hs.tris.size();
        uint32_t count = 0;
        for (uint32_t k = 0; k < g.count; ++k) { // the triangles of the leaf's references, each once; its record keeps the box of its pieces HERE
            const uint32_t ref = b.order[g.first + k], tri = b.ref_tri[ref];
            uint32_t at = count;
            for (uint32_t j = 0; j < count; ++j)
                if (hs.tris[first + j].global_index == tri) at = j;
            if (at == count) {
                hs.tris.push_back(gtris[tri]);
                for (int a = 0; a < 3; ++a) hs.rec_lo.push_back(b.tlo[3 * ref + a]), hs.rec_hi.push_back(b.thi[3 * ref + a]);
                ++count;
            } else {
                for (int a = 0; a < 3; ++a) {
                    float& lo = hs.rec_lo[3 * (size_t)(first + at) + a];
                    float& hi = hs.rec_hi[3 * (size_t)(first + at) + a];
                    lo = std::min(lo, b.tlo[3 * ref + a]), hi = std::max(hi, b.thi[3 * ref + a]);
                }
            }
        }
        hs.max_leaf_tris = std::max(hs.max_leaf_tris, count);
        hs.max_depth = std::max(hs.max_depth, depth);
        return ~(int32_t)((first << 2) | (count - 1));
    };
    hs.rec_lo.clear(), hs.rec_hi.clear();
    hs.n_split_triangles = b.n_split;
    const double root_area = std::max(1e-30, (double)half_area(g0.lo, g0.hi));
    double cost = 0.0;
    struct Item { int32_t gnode; int32_t slot; uint32_t depth; };
    std::deque<Item> q;
    hs.nodes.push_back(empty_node());
    if (g0.left < 0) { // a single leaf: give the root one leaf child
        if (g0.count > (uint32_t)kMaxLeafTris) { err = "internal: oversized root leaf"; return RT_ERR_INVALID; }
        BvhNode root = empty_node();
        float klo[1][3], khi[1][3];
        for (int a = 0; a < 3; ++a) klo[0][a] = g0.lo[a] - hs.pad, khi[0][a] = g0.hi[a] + hs.pad;
        if (!quantise_node(root, 1, klo, khi)) { err = "internal: quantisation failed"; return RT_ERR_INVALID; }
        root.child[0] = leaf_code(g0, 1);
        hs.nodes[0] = root;
        hs.sah_cost = (double)g0.count;
        return RT_OK;
    }
    // Emission order: largest surface area first until the LDS-staged top of the tree is complete (the first kTopNodes nodes of the
    // array are the children of the nodes a ray is most likely to visit; level order staged 2 % fewer expected visits), depth-first
    // below it, so that the nodes and the triangles of one subtree are contiguous in memory (fewer distinct cache lines per ray than
    // a pure breadth-first layout).
    // Which binary nodes become BVH4 nodes: dynamic programme over the binary tree (after Ylitie, Karras, Laine 2017).
    // best[n][k] = least SAH cost of covering subtree n with at most k child slots of its BVH4 parent; a subtree that
    // takes one slot is a leaf or a BVH4 node of its own (area + the best way to hand its 4 slots to its two halves).
    // cut[n][4] = slots a BVH4 node n gives to its LEFT half. RT_BVH_COLLAPSE=greedy restores the open-the-largest-child rule.
    bool use_dp = true;
    if (const char* e = dev_knob("RT_BVH_COLLAPSE")) use_dp = std::string(e) != "greedy";
    std::vector<std::array<float, 5>> best;
    std::vector<std::array<uint8_t, 5>> cut;
    if (use_dp) collapse_dp(b, best, cut);
    // children of BVH4 node: expand the recorded decisions (subtree n may use up to k slots)
    std::function<void(int32_t, int, int32_t*, int&)> gather = [&](int32_t n, int k, int32_t* kids, int& nk) {
        const GNode& g = b.nodes[n];
        if (k <= 1 || g.left < 0) { kids[nk++] = n; return; }
        // as a PARENT's subtree with k slots: either distribute them or fall back to k-1
        float dist = std::numeric_limits<float>::infinity();
        int dl = 1;
        for (int l = 1; l < k; ++l) {
            const float c = best[g.left][l] + best[g.right][k - l];
            if (c < dist) dist = c, dl = l;
        }
        if (dist < best[n][k - 1]) gather(g.left, dl, kids, nk), gather(g.right, k - dl, kids, nk);
        else gather(n, k - 1, kids, nk);
    };
    constexpr size_t kBfsNodes = kTopNodes; // the nodes the traversal kernels stage in LDS
    q.push_back({0, 0, 0});
    while (!q.empty()) {
        Item it;
        if (hs.nodes.size() < kBfsNodes) {
            size_t pick = 0; // the pending node a ray is most likely to visit: the one with the largest surface area
            float best_area = -1.0f;
            for (size_t i = 0; i < q.size(); ++i) {
                const float ar = half_area(b.nodes[q[i].gnode].lo, b.nodes[q[i].gnode].hi);
                if (ar > best_area) best_area = ar, pick = i;
            }
            it = q[pick];
            q.erase(q.begin() + (std::ptrdiff_t)pick);
        } else { it = q.back(); q.pop_back(); }
        const GNode& g = b.nodes[it.gnode];
        cost += (double)half_area(g.lo, g.hi) / root_area;
        int32_t kids[4] = {g.left, g.right, -1, -1};
        int nk = 2;
        if (use_dp) {
            nk = 0;
            const int li = cut[it.gnode][4];
            gather(g.left, li, kids, nk), gather(g.right, 4 - li, kids, nk);
        } else {
            while (nk < 4) { // greedy: open the inner child with the largest surface area
                int best = -1;
                float best_area = -1.0f;
                for (int k = 0; k < nk; ++k) {
                    const GNode& c = b.nodes[kids[k]];
                    if (c.left >= 0) {
                        float ar = half_area(c.lo, c.hi);
                        if (ar > best_area) best_area = ar, best = k;
                    }
                }
                if (best < 0) break;
                const GNode& c = b.nodes[kids[best]];
                kids[best] = c.left;
                kids[nk++] = c.right;
            }
        }
        BvhNode out = empty_node();
        Item pending[4];
        int n_pending = 0;
        float klo[4][3], khi[4][3];
        for (int k = 0; k < nk; ++k)
            for (int a = 0; a < 3; ++a) klo[k][a] = b.nodes[kids[k]].lo[a] - hs.pad, khi[k][a] = b.nodes[kids[k]].hi[a] + hs.pad;
        if (!quantise_node(out, nk, klo, khi)) { err = "internal: quantisation failed"; return RT_ERR_INVALID; }
        for (int k = 0; k < nk; ++k) {
            const GNode& c = b.nodes[kids[k]];
            if (c.left < 0) {
                if (c.count == 0 || c.count > (uint32_t)kMaxLeafTris) { err = "internal: bad leaf size"; return RT_ERR_INVALID; }
                out.child[k] = leaf_code(c, it.depth + 1);
                cost += (double)half_area(c.lo, c.hi) / root_area * (double)c.count;
            } else {
                out.child[k] = (int32_t)hs.nodes.size();
                hs.nodes.push_back(empty_node());
                pending[n_pending++] = {kids[k], out.child[k], it.depth + 1};
            }
        }
        // top phase: children in order at the back; depth-first phase: reversed, so child 0 is next
        if (hs.nodes.size() < kBfsNodes) for (int k = 0; k < n_pending; ++k) q.push_back(pending[k]);
        else for (int k = n_pending - 1; k >= 0; --k) q.push_back(pending[k]);
        hs.nodes[it.slot] = out;
    }
    hs.sah_cost = cost;
    if (const char* e = dev_knob("RT_BVH_VERBOSE")) { // diagnostic: leaves by size
        if (e[0] == '1') {
            size_t by[5] = {0, 0, 0, 0, 0};
            for (const BvhNode& n : hs.nodes)
                for (int k = 0; k < 4; ++k)
                    if (n.child[k] < 0 && n.child[k] != kChildEmpty) by[(((uint32_t)~n.child[k]) & 3u) + 1u]++;
            std::fprintf(stderr, "[rt bvh] %zu nodes, leaves with 1 / 2 / 3 / 4 triangles: %zu / %zu / %zu / %zu, %zu records for %u triangles (%u split)\n", hs.nodes.size(),
                         by[1], by[2], by[3], by[4], hs.tris.size(), T, hs.n_split_triangles);
        }
    }
    if (hs.tris.size() < T || hs.tris.size() > b.order.size()) { err = "internal: leaf record count mismatch after build"; return RT_ERR_INVALID; }
    hs.stack_need = need(0);
    if (hs.stack_need + 2 >= (uint32_t)kStackSize) {
        if (bvh_kind != RT_BVH_MEDIAN_INTERNAL) return build_host_scene(d, RT_BVH_MEDIAN_INTERNAL, hs, err); // balanced fallback
        err = "internal: BVH deeper than the traversal stack";
        return RT_ERR_INVALID;
    }
    return RT_OK;
}

int check_bvh(const HostScene& hs, std::string& err) {
    const size_t T = hs.wverts.size() / 9;
    if (hs.nodes.empty()) { err = "no root node"; return RT_ERR_INVALID; }
    std::vector<uint32_t> seen(T, 0);
    // pre-split trees (scene_build.cpp: presplit): a triangle may sit in several leaves, each with the box of its pieces there
    const bool split = hs.rec_lo.size() == 3 * hs.tris.size() && hs.rec_hi.size() == 3 * hs.tris.size() && hs.n_split_triangles > 0;
    std::vector<std::vector<uint32_t>> pieces(split ? T : 0);
    size_t visited = 0;
    bool ok = true;
    const float inf = std::numeric_limits<float>::infinity();
    // returns the exact bounds of everything below `child`; checks them against every decoded box above
    struct B { float lo[3], hi[3]; };
    auto fail = [&](const std::string& m) { if (ok) err = m; ok = false; };
    std::function<B(int32_t, uint32_t)> walk = [&](int32_t child, uint32_t depth) -> B {
        B r;
        for (int a = 0; a < 3; ++a) r.lo[a] = inf, r.hi[a] = -inf;
        if (!ok) return r;
        if (depth >= (uint32_t)kStackSize) { fail("tree deeper than traversal stack"); return r; }
        if (child >= 0) {
            if ((size_t)child >= hs.nodes.size()) { fail("child index out of range"); return r; }
            if (++visited > hs.nodes.size()) { fail("cycle in node graph"); return r; }
            const BvhNode& n = hs.nodes[child];
            for (int k = 0; k < 4; ++k) {
                if (n.child[k] == kChildEmpty) continue;
                B c = walk(n.child[k], depth + 1);
                if (!ok) return r;
                for (int a = 0; a < 3; ++a) {
                    const float s = n.scale(a);
                    const float dlo = n.origin[a] + (float)((n.q[2 * a] >> (8 * k)) & 0xffu) * s;
                    const float dhi = n.origin[a] + (float)((n.q[2 * a + 1] >> (8 * k)) & 0xffu) * s;
                    // the decoded box must contain the subtree with at least half the builder's pad to spare
                    if (!(dlo <= c.lo[a] - 0.5f * hs.pad) || !(dhi >= c.hi[a] + 0.5f * hs.pad)) {
                        fail("decoded child box does not contain its subtree (node " + std::to_string(child) + ")");
                        return r;
                    }
                    r.lo[a] = std::min(r.lo[a], c.lo[a]), r.hi[a] = std::max(r.hi[a], c.hi[a]);
                }
            }
        } else {
            uint32_t code = (uint32_t)~child;
            uint32_t first = code >> 2, count = (code & 3u) + 1u;
            if ((size_t)first + count > hs.tris.size()) { fail("leaf range out of bounds"); return r; }
            for (uint32_t k = 0; k < count; ++k) {
                const TriRec& tr = hs.tris[first + k];
                if (tr.global_index >= T) { fail("leaf triangle has bad global index"); return r; }
                for (uint32_t j = 0; j < k; ++j)
                    if (hs.tris[first + j].global_index == tr.global_index) { fail("triangle twice in one leaf"); return r; }
                if (seen[tr.global_index]++ && !split) { fail("triangle appears in two leaves"); return r; }
                const float* w = &hs.wverts[9 * (size_t)tr.global_index];
                if (tr.v0[0] != w[0] || tr.v0[1] != w[1] || tr.v0[2] != w[2]) { fail("leaf triangle data mismatch"); return r; }
                if (split) { // the box of the triangle's pieces in THIS leaf (the whole triangle where it was not split)
                    const float* plo = &hs.rec_lo[3 * (size_t)(first + k)];
                    const float* phi = &hs.rec_hi[3 * (size_t)(first + k)];
                    for (int a = 0; a < 3; ++a) r.lo[a] = std::min(r.lo[a], plo[a]), r.hi[a] = std::max(r.hi[a], phi[a]);
                    pieces[tr.global_index].push_back(first + k);
                } else {
                    for (int v = 0; v < 3; ++v)
                        for (int a = 0; a < 3; ++a) r.lo[a] = std::min(r.lo[a], w[3 * v + a]), r.hi[a] = std::max(r.hi[a], w[3 * v + a]);
                }
            }
        }
        return r;
    };
    (void)walk(0, 0);
    if (!ok) return RT_ERR_INVALID;
    for (size_t t = 0; t < T; ++t)
        if (!seen[t]) { err = "triangle " + std::to_string(t) + " is in no leaf"; return RT_ERR_INVALID; }
    if (split) { // the pieces of a split triangle must cover it: every point of a barycentric grid lies in the box of one of its records
        for (size_t t = 0; t < T; ++t) {
            if (pieces[t].size() < 2) continue;
            const float* w = &hs.wverts[9 * t];
            for (int i = 0; i <= 8; ++i)
                for (int j = 0; i + j <= 8; ++j) {
                    const double u = i / 8.0, v = j / 8.0, ww = 1.0 - u - v;
                    double p[3];
                    for (int a = 0; a < 3; ++a) p[a] = ww * w[a] + u * w[3 + a] + v * w[6 + a];
                    bool in = false;
                    for (uint32_t rec : pieces[t]) {
                        bool inside = true;
                        for (int a = 0; a < 3; ++a) {
                            const double tol = 1e-6 * (std::fabs(p[a]) + 1.0);
                            inside = inside && p[a] >= (double)hs.rec_lo[3 * (size_t)rec + a] - tol && p[a] <= (double)hs.rec_hi[3 * (size_t)rec + a] + tol;
                        }
                        in = in || inside;
                    }
                    if (!in) { err = "the pieces of split triangle " + std::to_string(t) + " do not cover it"; return RT_ERR_INVALID; }
                }
        }
    }
    return RT_OK;
}

} // namespace rt

// ---- diagnostic: node visits of a closest-hit walk on the host (rt_scene_count_visits) --------------------------------------------------------
// The walk the traversal kernels make — children entered nearest first, the others stacked, everything beyond the closest hit so far culled —
// with a choice of the child boxes it tests:
//   mode 0  the decoded quantised boxes, as the kernels test them
//   mode 1  the exact bounds of each child's subtree, padded as the builder pads them: what an unquantised (fp32) node would hold
//   mode 2  the exact padded bounds re-quantised on the node's grid with 2 more bits per plane (step / 4)
// so that the share of the visits that is the price of 8-bit planes can be read off. Counts only: no result of a
// render depends on it.
namespace rt {
int count_visits(const HostScene& hs, uint32_t n, const float* org, const float* dir, int mode, uint64_t* node_visits, uint64_t* tri_tests, float* t_out,
                 uint32_t* tri_out, std::string& err) {
    if (mode < 0 || mode > 2) { err = "mode: 0 quantised, 1 exact, 2 quantised with two more bits"; return RT_ERR_INVALID; }
    if (hs.nodes.empty()) { err = "no tree"; return RT_ERR_INVALID; }
    const float inf = std::numeric_limits<float>::infinity();
    const bool split = hs.rec_lo.size() == 3 * hs.tris.size() && hs.rec_hi.size() == 3 * hs.tris.size() && hs.n_split_triangles > 0;
    struct B { float lo[3], hi[3]; };
    std::vector<B> box(hs.nodes.size() * 4); // the box tested for child k of node i
    std::function<B(int32_t)> walk = [&](int32_t child) -> B {
        B r;
        for (int a = 0; a < 3; ++a) r.lo[a] = inf, r.hi[a] = -inf;
        if (child >= 0) {
            const BvhNode& nd = hs.nodes[(size_t)child];
            for (int k = 0; k < 4; ++k) {
                if (nd.child[k] == kChildEmpty) continue;
                const B c = walk(nd.child[k]);
                B& out = box[(size_t)child * 4 + k];
                for (int a = 0; a < 3; ++a) {
                    r.lo[a] = std::min(r.lo[a], c.lo[a]), r.hi[a] = std::max(r.hi[a], c.hi[a]);
                    const float s = nd.scale(a);
                    if (mode == 0) {
                        out.lo[a] = nd.origin[a] + (float)((nd.q[2 * a] >> (8 * k)) & 0xffu) * s;
                        out.hi[a] = nd.origin[a] + (float)((nd.q[2 * a + 1] >> (8 * k)) & 0xffu) * s;
                    } else if (mode == 1) {
                        out.lo[a] = c.lo[a] - hs.pad, out.hi[a] = c.hi[a] + hs.pad;
                    } else {
                        const float q = s * 0.25f;
                        out.lo[a] = nd.origin[a] + std::floor((c.lo[a] - hs.pad - nd.origin[a]) / q) * q;
                        out.hi[a] = nd.origin[a] + std::ceil((c.hi[a] + hs.pad - nd.origin[a]) / q) * q;
                    }
                }
            }
        } else {
            const uint32_t code = (uint32_t)~child, first = code >> 2, count = (code & 3u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
                if (split) {
                    for (int a = 0; a < 3; ++a)
                        r.lo[a] = std::min(r.lo[a], hs.rec_lo[3 * (size_t)(first + k) + a]), r.hi[a] = std::max(r.hi[a], hs.rec_hi[3 * (size_t)(first + k) + a]);
                } else {
                    const float* w = &hs.wverts[9 * (size_t)hs.tris[first + k].global_index];
                    for (int v = 0; v < 3; ++v)
                        for (int a = 0; a < 3; ++a) r.lo[a] = std::min(r.lo[a], w[3 * v + a]), r.hi[a] = std::max(r.hi[a], w[3 * v + a]);
                }
            }
        }
        return r;
    };
    (void)walk(0);
    uint64_t visits = 0, tests = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const float o[3] = {org[3 * i], org[3 * i + 1], org[3 * i + 2]}, d[3] = {dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]};
        float inv[3];
        for (int a = 0; a < 3; ++a) inv[a] = 1.0f / (std::fabs(d[a]) < 1e-30f ? std::copysign(1e-30f, d[a]) : d[a]);
        float best = inf;
        uint32_t best_tri = kNoTri;
        int32_t stack[kStackSize * 4];
        int sp = 0;
        int32_t cur = 0;
        for (;;) {
            if (cur >= 0) {
                visits++;
                const BvhNode& nd = hs.nodes[(size_t)cur];
                float key[4];
                int32_t ch[4];
                int m = 0;
                for (int k = 0; k < 4; ++k) {
                    if (nd.child[k] == kChildEmpty) continue;
                    const B& b = box[(size_t)cur * 4 + k];
                    float tn = 0.0f, tf = best;
                    for (int a = 0; a < 3; ++a) {
                        const float t0 = (b.lo[a] - o[a]) * inv[a], t1 = (b.hi[a] - o[a]) * inv[a];
                        tn = std::max(tn, std::min(t0, t1)), tf = std::min(tf, std::max(t0, t1));
                    }
                    if (tn <= tf) key[m] = tn, ch[m] = nd.child[k], m++;
                }
                for (int a = 1; a < m; ++a) // nearest first
                    for (int b = a; b > 0 && key[b] < key[b - 1]; --b) std::swap(key[b], key[b - 1]), std::swap(ch[b], ch[b - 1]);
                for (int a = m - 1; a >= 1; --a) stack[sp++] = ch[a];
                if (m) { cur = ch[0]; continue; }
            } else {
                const uint32_t code = (uint32_t)~cur, first = code >> 2, count = (code & 3u) + 1u;
                for (uint32_t k = 0; k < count; ++k) {
                    tests++;
                    const TriRec& tr = hs.tris[first + k];
                    // Moller-Trumbore on (v0, e1, e2), in double: only the count matters here
                    const double e1[3] = {tr.e1[0], tr.e1[1], tr.e1[2]}, e2[3] = {tr.e2[0], tr.e2[1], tr.e2[2]};
                    const double p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
                    const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
                    if (det == 0.0) continue;
                    const double tv[3] = {o[0] - tr.v0[0], o[1] - tr.v0[1], o[2] - tr.v0[2]};
                    const double u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) / det;
                    const double q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                    const double v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) / det;
                    if (u < 0.0 || v < 0.0 || u + v > 1.0) continue;
                    const double t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) / det;
                    if (t > (double)kTNear && (t < best || (t == best && tr.global_index < best_tri))) best = (float)t, best_tri = tr.global_index;
                }
            }
            if (sp == 0) break;
            cur = stack[--sp];
        }
        if (t_out) t_out[i] = best;
        if (tri_out) tri_out[i] = best_tri;
    }
    if (node_visits) *node_visits = visits;
    if (tri_tests) *tri_tests = tests;
    return RT_OK;
}
} // namespace rt
