// rt_internal.h — what the translation units of librt_mi355x.so share: error plumbing, the scene and renderer objects behind the opaque handles
// of include/rt_mi355x.h, and the frame machinery's entry points.
//   rt_abi.hip     the C entry points for scenes, renderers and frames (argument checks, object life time)
//   rt_frame.hip   tile buffers, schedules, the launches of a frame (the only unit that includes the render kernels, rt_kernels.h)
//   rt_comm.hip    the multi-GPU frame gather (RCCL) and its de-interleave kernel
//   rt_probes.hip  rt_intersect_batch and the rt_probe_* entry points with their kernels (rt_probe_kernels.h)
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_knobs.h"
#include "rt_launch.h"
#include "scene_build.h"

namespace rtlib {

extern thread_local std::string g_err; // defined in rt_abi.hip (rt_last_error)

inline int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP,                           \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                            \
    } while (0)

// nothing may cross the C ABI as a C++ exception (std::vector growth inside the host orchestration can throw)
template <typename F>
int no_throw(F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_OOM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(RT_ERR_INVALID, std::string("internal error: ") + e.what());
    }
}

inline int device_ok(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device available (the render path has no CPU fallback)");
    if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    return RT_OK;
}

// device allocation that frees itself on every return path of the C entry points
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

template <typename T>
int upload(const std::vector<T>& v, const T** out, uint64_t& bytes) {
    *out = nullptr;
    const size_t n = std::max<size_t>(v.size(), 1);
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, n * sizeof(T)));
    *out = (const T*)p; // owned by the scene from here on: rt_scene_destroy frees it even if the copy fails
    bytes += n * sizeof(T);
    if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

} // namespace rtlib

using namespace rt;
using namespace rtlib;

struct rt_scene {
    HostScene hs;
    int device = -1;
    SceneDev dev{};
    uint64_t device_bytes = 0;
};

// kernel launches of one frame, in all and per kernel family (rt_stats::launches_by_kernel)
struct LaunchCount {
    uint32_t total = 0;
    uint32_t by[RT_K_COUNT] = {};
    void add(int kernel, uint32_t n = 1) { total += n, by[kernel] += n; }
    void also(int kernel, uint32_t n = 1) { by[kernel] += n; } // a second family of a launch already counted
};

struct rt_renderer {
    int kind = 0;
    rt_scene* scene = nullptr;
    int32_t width = 0, height = 0;
    uint32_t max_depth = 0, spp = 0, seed_mode = 0;
    TileDev tile{0, 1, 8, 0};
    uint32_t n_local = 0; // pixels in this tile
    bool unusable = false; // the last (re-)allocation of the tile's buffers failed: no frame until one succeeds
    int n_cus = 256;      // compute units of the device (persistent-kernel grid sizing)
    hipStream_t stream = nullptr;
    // outputs (device) used when the caller passes host pointers
    float* d_f32 = nullptr;
    uint8_t* d_u8 = nullptr;
    // megakernel
    unsigned long long* d_ray_counter = nullptr;
    void* d_slice_state = nullptr;         // pixel slices (rt_kernels.h: SliceDev): 32 bytes per pixel of the tile, zero when allocated, never cleared
    MegaFrame* d_frame = nullptr;          // the frame's constants k_megakernel reads from device memory: camera + slices (copied per frame from h_frame, pinned)
    MegaFrame* h_frame = nullptr;
    int stats_level = 2;                   // RT_KERNEL_STATS: 2 = the full counters (=1), 1 = the megakernel's timing-only instantiation (=2)
    uint32_t last_slices = 1;              // slices the wavefront renderer's last enqueued frame used (rt_stats.pixel_slices)
    uint32_t slice_tag = 0;                // tag0 of the last frame (grows by kMaxSlices per frame)
    uint32_t rr_start = 0;                 // Russian roulette from this bounce on (0 = off, the reference's behaviour)
    unsigned long long* d_stats = nullptr; // RT_KERNEL_STATS=1: wave scheduling statistics (diagnostic)
    // wavefront: the tile is rendered as K interleaved sub-tiles ("lanes"), each with its own queues and
    // its own HIP stream. The lanes' launch chains are independent, so the drain tail of one lane's EXTEND
    // launch and its bandwidth-bound SHADE launch overlap with the other lane's traversal
    // (measured: 1 lane 3.85, 2 lanes 4.57, 3 lanes 4.63 Grays/s).
    struct WfLane {
        TileDev tile{0, 1, 8, 0};
        uint32_t n_local = 0;
        QueueDev q[2]{};
        uint32_t* d_rng = nullptr;
        float4* d_accum = nullptr;
        uint32_t* d_counts = nullptr; // spp * (max_depth + 1) queue lengths, then spp * max_depth extend cursors
        float4* d_hits = nullptr;     // closest-hit record per queue slot
        // the streaming schedule's dynamic queue (k_wf_finish<.., true>): entries (all ones before every launch) and control words
        // (tail, head: one 64-byte line each, zeroed before every launch)
        unsigned long long* d_rq_entries = nullptr;
        uint32_t* d_rq_ctl = nullptr;
        uint32_t rq_capacity = 0;
        // cost ordering of the one-launch schedule (k_wf_tile_cost / k_wf_order_tiles): rays of every pixel's first sample, the 8x8
        // blocks' keys and their order
        uint8_t* d_cost = nullptr;
        uint32_t* d_tile_key = nullptr;
        uint32_t* d_tile_order = nullptr;
        uint32_t n_tiles = 0; // 0: no cost ordering for this tile
        std::vector<uint32_t> h_counts;
        unsigned long long* d_extra = nullptr; // per launch: rays k_wf_finish traced beyond its queue's length; [spp]: the dynamic queue's error word
        std::vector<unsigned long long> h_extra;
        hipStream_t stream = nullptr; // from the device's lane-stream pool (lane_stream_of) unless stream_owned
        bool stream_owned = false;
        hipEvent_t ev_done = nullptr;
        std::vector<hipEvent_t> ev_pool;
        size_t n_hot = 0;
        struct Span { uint32_t e0, e1; int kernel; }; // profiling: ev_pool[e0] .. ev_pool[e1] bracket one launch of `kernel`
        std::vector<Span> spans;
    };
    std::vector<WfLane> lanes;
    hipEvent_t ev_fork = nullptr;
    // The wavefront frame is a fixed sequence of ~2 * spp * depth dependent launches on the lanes' streams (queue
    // lengths stay on the device), so it CAN be captured once into a hipGraph and replayed per frame; the camera, the
    // only per-frame input, is read from d_cam, refreshed by a copy node from the pinned h_cam. Opt-in (RT_WF_GRAPH=1):
    // on ROCm 7.2 the replayed graph is 1.7x SLOWER than the same launches issued on the two streams (atrium 1080p,
    // 8 spp: 49.2 ms vs 29.6 ms; the stream launches are already queued far ahead of the device, and the graph's
    // two branches no longer overlap), so plain launches stay the default.
    CameraDev* d_cam = nullptr;
    CameraDev* h_cam = nullptr;
    // The wavefront renderer's schedule (include/rt_mi355x.h: rt_schedule), set through rt_renderer_set_schedule; the RT_WF_*
    // environment variables only fill the initial value at rt_renderer_create. What ran is reported in rt_stats.
    //   finish_depth: bounces rendered launch by launch (EXTEND + SHADE) before k_wf_finish follows the surviving rays to the end of
    //   their paths; >= max_depth = the reference's schedule (a launch pair per bounce).
    //   reorder / matsort (f-3): SHADE's compaction bins a block's survivors by (direction octant, 4x4x4 cell of the
    //   origin) / SHADE partitions a block's rays by material kind before shading them (measured: profiles/r02_reorder.json)
    rt_schedule sched{0u, 0u, 0u, -1, 0u, 0u, -1, 0u, 0u, -1};
    // waves per SIMD one k_wf_extend launch asks for; 0 = all 8 on one stream lane, 4 (one 1024-thread workgroup per CU) where lanes overlap: the
    // lanes' launches then sit side by side instead of queueing workgroups behind each other (3 lanes: 190.4 -> 186.7 ms per frame, 2 lanes 205.0 ->
    // 199.0; 2 / 6 waves: 201.6 / 190.8). RT_WF_EXTEND_OCC, tuning
    uint32_t wf_extend_occ = 0;
    uint32_t wf_finish_occ = 0; // waves per SIMD one k_wf_finish launch asks for; 0 = kMegaWaves / stream lanes (RT_WF_FINISH_OCC, tuning)
    // SHOOT's input claims (round 3). A device atomic on the queue cursor serialises (~6 ns each, more under contention): with 128-slot
    // claims a 2 M-ray launch made 16 k of them, and smaller claims were far worse (64 slots: 4.7 -> 3.8 Grays/s, 32: 2.3). Now 85 % of
    // the queue is handed out STATICALLY in equal parts (whole 64-slot generations per wave) and only the rest goes through the cursor,
    // in claims of a share of what is left (at most 256 slots, at least the wave's idle lanes or 16) so that no wave starts a long
    // private chunk when the others are about to leave: 2 / 3 stream lanes 5.20 / 5.28 -> 5.66 / 5.75 Grays/s at 16 spp.
    uint32_t wf_shoot_chunk = 0; // queue slots per cursor claim at most; 0 = 256 guided, 128 (64 on small tiles) unguided (RT_WF_SHOOT_CHUNK)
    bool wf_shoot_tail = true;   // guided claims (RT_WF_SHOOT_TAIL=0: a fixed chunk per claim, the first one static, as EXTEND)
    uint32_t wf_shoot_static_pct = 85; // guided: the share of the queue handed out statically (RT_WF_SHOOT_STATIC_PCT)
    hipGraphExec_t graph_exec = nullptr;
    float* graph_f32 = nullptr; // outputs the captured graph writes
    uint8_t* graph_u8 = nullptr;
    hipStream_t graph_stream = nullptr;
    LaunchCount graph_launches{};
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // rt_frame_gather_begin reads the renderer's tile buffers (d_f32 / d_u8) asynchronously, ordered on the renderer's OWN stream. A next frame
    // begun on a caller's stream is not ordered behind that: ev_tile_read (recorded on the own stream behind the send / the copies) is what
    // frame_begin makes such a stream wait for before anything writes the tile again.
    hipEvent_t ev_tile_read = nullptr;
    bool tile_read_recorded = false;
    // the frame between rt_render_frame_begin and rt_render_frame_end
    bool frame_pending = false;
    hipStream_t pending_stream = nullptr;
    rt_stats pending_stats{};
    size_t pending_hot = 0;
    std::chrono::high_resolution_clock::time_point pending_t0{};
    bool profile_kernels = false;
    std::vector<hipEvent_t> ev_pool;
    // diagnostics: RT_MEGA_LDS_PAD=<bytes> of unused dynamic LDS per workgroup lowers the number of
    // resident workgroups per CU without touching the code; RT_MEGA_OCC=<waves per SIMD> sizes the persistent grid to match
    uint32_t hw_queues = 4; // GPU_MAX_HW_QUEUES as the host had set it when the renderer was created (4 = HIP's default): bounds the automatic stream lanes
    uint32_t mega_lds_pad = 0, mega_occ = kMegaWaves;
};

namespace rtlib {
// rt_frame.hip
void drop_graph(rt_renderer* r);
void free_tile_buffers(rt_renderer* r);
void drain_streams(rt_renderer* r, hipStream_t extra);
bool origin_in_contract_range(const HostScene& hs, const float o[3]);
uint32_t hw_queues_from_env();
hipError_t lane_stream_of(int device, uint32_t k, hipStream_t* out, bool* owned);
int alloc_tile_buffers(rt_renderer* r);
int frame_begin(rt_renderer* r, const rt_camera* cam, float* d_f32, uint8_t* d_u8, hipStream_t st);
int frame_end(rt_renderer* r, rt_stats* stats);
int render_impl(rt_renderer* r, const rt_camera* cam, float* d_f32, uint8_t* d_u8, hipStream_t st, rt_stats* stats);
} // namespace rtlib
