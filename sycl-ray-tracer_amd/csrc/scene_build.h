// scene_build.h — host side of rt_scene: flattening instances to world-space triangles and
// building the BVH (replaces Embree's rtcCommitScene: src/scene.cpp:101-107,406-439,487-507).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_types.h"

namespace rt {

struct HostScene {
    std::vector<BvhNode> nodes;   // node 0 is the root (always an inner node)
    std::vector<TriRec> tris;     // leaf order
    std::vector<ShadeRec> shade;  // global order
    std::vector<InstRec> inst;
    std::vector<MatRec> mats;
    std::vector<uint8_t> tex;
    std::vector<float> wverts;    // 9 floats per triangle, global order (diagnostics / checks)
    std::vector<float> rec_lo, rec_hi; // host SAH builder: per leaf record, the box of the triangle's pieces in that leaf (check_bvh)
    uint32_t n_split_triangles = 0;    // triangles the SAH builder's pre-splitting pass cut into several references
    uint32_t n_layers = 0;
    bool packed_mat = false;      // ShadeRec::instance = instance | material << 20 (rt_types.h)
    float sky[3] = {0.5f, 0.7f, 1.0f};
    float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};
    float pad = 0.0f;             // absolute box padding used by the builder
    uint32_t max_depth = 0, max_leaf_tris = 0, stack_need = 0;
    double sah_cost = 0.0;
};

constexpr int RT_BVH_MEDIAN_INTERNAL = 99; // balanced fallback when a tree would overflow the traversal stack

// Validates `desc` and fills `out`. Returns RT_OK or RT_ERR_INVALID with `err` set.
int build_host_scene(const rt_scene_desc* desc, int bvh_kind, HostScene& out, std::string& err);

// lbvh_gpu.hip: BVH construction on the current HIP device; fills hs.nodes / hs.tris (downloaded copies).
int build_lbvh_gpu(HostScene& hs, const std::vector<TriRec>& gtris, std::string& err);

// Structural check used by rt_scene_check_bvh.
int check_bvh(const HostScene& hs, std::string& err);

// Diagnostic used by rt_scene_count_visits: closest-hit walks on the host with quantised (0), exact (1) or finer quantised (2) child boxes.
int count_visits(const HostScene& hs, uint32_t n, const float* org, const float* dir, int mode, uint64_t* node_visits, uint64_t* tri_tests, float* t_out,
                 uint32_t* tri_out, std::string& err);

} // namespace rt
