// rt_mi355x.hip — implementation of the C ABI in include/rt_mi355x.h on HIP (gfx950).
//
// Host orchestration of the two renderers. Unlike the reference, which blocks on the host after
// every kernel (769 blocking submits per 64-spp frame: src/render_wavefront.cpp:396-431), a frame
// here is ONE asynchronous stream of launches: queue lengths live in device memory, one counter
// per (sample, depth), so no bounce ever waits for the host.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: librccl.so is dlopen()ed by rt_comm_create
#include <dlfcn.h>

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
#include "rt_kernels.h"
#include "scene_build.h"
#include "rt_knobs.h"

using namespace rt;

// Schedule of the wavefront renderer (measured, atrium 1080p 8 spp, Mrays/s): every bounce as its own EXTEND + SHADE launch pair (the
// reference's schedule) 5103; first 1 / 2 / 3 / 4 / 6 bounces that way, the rest of each path in k_wf_finish 4923 / 4772 / 4942 / 4873 / 5028;
// everything in k_wf_finish (one compacted queue per sample: generate -> queue -> persistent waves) 5282 when each of the two
// overlapping launches asks for all 6 wave slots per SIMD, 6086 when each asks for its half. BASELINE config 5's tile (4K, depth 16):
// 56.8 ms -> 39.2 ms against the megakernel's 30.9. Default: 0 bounces launch by launch.
// Samples k_wf_finish renders per pixel and launch (RT_WF_SAMPLES_PER_LAUNCH). 0, the default, is the ONE-LAUNCH schedule: all samples in
// one launch per frame and tile on one stream lane, the lane keeps its pixel (depth first) and a wave takes exactly as many queue slots as
// it has idle lanes — the megakernel's loop fed from a queue, and as fast: atrium 1080p 64 spp 151.7 ms (megakernel 149.8), 1/2 / 1/4 /
// 1/8 of it 82.0 / 51.0 / 29.4 ms (81.5 / 50.9 / 29.6), config 5's tile 462 ms (454), the full 4K frame at 16 spp 142.5 ms (140.8).
// A positive value selects the earlier schedules, all bit-exact, on two stream lanes (RT_WF_STREAMS):
//  - 1: a launch per sample (the reference's shape minus its per-bounce launches): 167.9 ms at 1080p 64 spp; every launch ends with its
//    slowest paths while its half of the machine idles;
//  - C > 1, breadth first (RT_WF_REQUEUE=1, default): a pixel between two samples goes through a device-wide dynamic queue to whichever
//    lane is idle. Depth 10, 16 spp, Mrays/s at C = 1 / 4 / 16: 960x540 4253 / 4244 / 4368; 1920x540 5081 / 5830 / 5869; 1920x1080 6334 /
//    6179 / 5946; 3840x2160 7032 / 6248 / 5843: it levels off at ~6.0 Grays/s (+12 % VALU instructions, +32 % L2 misses: a wave's pixels
//    are no longer neighbours) and passes ~1.3 G pixels per second at most (Cornell box: 36.2 ms a launch per sample, 50.9 ms streaming);
//    it wins over a launch per sample on small tiles and deep paths (config 5's tile 629 -> 495 ms; 1080p depth 50: 5708 -> 6486 Mrays/s);
//  - C > 1, depth first on two stream lanes (RT_WF_REQUEUE=0): 6323 / 6148 / 6012 / 5826 / 5906 Mrays/s for C = 1 / 2 / 4 / 8 / 16 at 1080p
//    16 spp when waves took private 128-slot chunks (whole pixel-chains: the launch ended with a few waves working through theirs).

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
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

int device_ok(int device) {
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

} // namespace

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
    u32x4* d_slice_state = nullptr;        // pixel slices (rt_kernels.h: SliceDev): 32 bytes per pixel of the tile, zero when allocated, never cleared
    MegaFrame* d_frame = nullptr;          // the frame's constants k_megakernel reads from device memory: camera + slices (copied per frame from h_frame, pinned)
    MegaFrame* h_frame = nullptr;
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

namespace {

void drop_graph(rt_renderer* r) {
    if (r->graph_exec) (void)hipGraphExecDestroy(r->graph_exec);
    r->graph_exec = nullptr;
}

void free_tile_buffers(rt_renderer* r) {
    drop_graph(r);
    auto fr = [](auto*& p) {
        if (p) (void)hipFree((void*)p);
        p = nullptr;
    };
    fr(r->d_f32), fr(r->d_u8), fr(r->d_ray_counter), fr(r->d_stats), fr(r->d_slice_state);
    for (auto& L : r->lanes) {
        fr(L.d_rng), fr(L.d_accum), fr(L.d_counts), fr(L.d_hits), fr(L.d_rq_entries), fr(L.d_rq_ctl), fr(L.d_extra), fr(L.d_cost), fr(L.d_tile_key), fr(L.d_tile_order);
        for (auto& q : L.q) fr(q.org_id), fr(q.a), fr(q.b), fr(q.c);
        for (hipEvent_t e : L.ev_pool) (void)hipEventDestroy(e);
        if (L.ev_done) (void)hipEventDestroy(L.ev_done);
        if (L.stream && L.stream_owned) (void)hipStreamDestroy(L.stream);
    }
    r->lanes.clear();
}

// waits for everything queued on the renderer's streams (its own, the wavefront lanes', and `extra` if given)
void drain_streams(rt_renderer* r, hipStream_t extra) {
    if (extra) (void)hipStreamSynchronize(extra);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    for (auto& L : r->lanes)
        if (L.stream) (void)hipStreamSynchronize(L.stream);
}

// Range of the closest-hit contract (include/rt_mi355x.h, rt_intersect_batch): a ray origin may lie at most kContractRange scene
// scales outside the scene's bounds on any axis, the scene scale being what the padding of the BVH boxes is derived from
// (scene_build.cpp: max(largest extent, largest |coordinate|)). Farther out the fp32 rounding of the ray itself exceeds the padding
// and the box test could cull a hit the triangle test would report; the entry points refuse such origins instead of answering
// silently wrong. NaN origins are refused too.
constexpr float kContractRange = 100.0f;
bool origin_in_contract_range(const HostScene& hs, const float o[3]) {
    if (hs.wverts.empty()) return std::isfinite(o[0]) && std::isfinite(o[1]) && std::isfinite(o[2]); // nothing to cull
    float scale = 0.0f;
    for (int a = 0; a < 3; ++a)
        scale = std::max(scale, std::max(hs.bounds_hi[a] - hs.bounds_lo[a], std::max(std::fabs(hs.bounds_lo[a]), std::fabs(hs.bounds_hi[a]))));
    // a degenerate scene (all geometry in the origin: scale 0) has nothing a box test could cull wrongly: every finite origin is in range
    if (!(scale > 0.0f)) return std::isfinite(o[0]) && std::isfinite(o[1]) && std::isfinite(o[2]);
    const float limit = kContractRange * scale;
    for (int a = 0; a < 3; ++a) {
        const float outside = std::max(std::max(hs.bounds_lo[a] - o[a], o[a] - hs.bounds_hi[a]), 0.0f);
        if (!(outside <= limit)) return false;
    }
    return true;
}

int32_t rows_of_tile(const TileDev& t, int32_t height) {
    int32_t rows = 0;
    for (int32_t y = 0; y < height; ++y)
        if (((uint32_t)y / t.strip_rows) % t.world == t.rank) rows++;
    return rows;
}

// The wavefront renderer's default schedule: ONE k_wf_generate + ONE k_wf_finish launch per frame, tile and stream lane, every
// pixel kept in its lane for all its samples (rt_kernels.h): rt_schedule{finish_depth 0, samples_per_launch 0}.
bool wf_one_launch(const rt_renderer* r) { return r->sched.samples_per_launch == 0 && r->sched.finish_depth == 0 && r->max_depth > 0; }
// Stream lanes: interleaved sub-tiles of the renderer's tile, each with its own queues on a HIP stream of its own, so that one lane's launch
// tails overlap with another lane's traversal. 1 for the one-launch schedule; where every bounce is a launch (pair) 3 pay (split EXTEND + SHADE on
// 1 / 2 / 3 lanes: 4166 / 4979 / 5213 Mrays/s; fused SHOOT on four lanes leaves each launch one wave per SIMD and loses a fifth); 2 otherwise.
// HIP maps a process's streams onto at most GPU_MAX_HW_QUEUES hardware queues — FOUR unless the host sets the variable before its first HIP call.
// The frame's stream, three lane streams and the host framework's own stream are five: two lanes then share a queue and their launches run one
// after the other (measured: the split pipeline on three lanes 314 ms per frame with 4 queues, 191 ms with 8). The library never touches the
// process environment (round 3 set the variable from a constructor: a side effect of dlopen, and void when HIP was already up). It READS what the
// host configured when a renderer is created, and resolves an automatic lane count down to what the queues can serve: lanes + 2 <= queues. The
// host (bench.py, the raytracer CLI) sets GPU_MAX_HW_QUEUES=8 itself before HIP starts; rt_stats reports both numbers.
uint32_t hw_queues_from_env() { // read ONCE per process, at the library's first renderer: HIP latched the value at its own initialisation, and a
    static const uint32_t queues = [] { // host that changes the variable later must not make the library report queues HIP does not serve
        const char* e = std::getenv("GPU_MAX_HW_QUEUES");
        const int v = e ? std::atoi(e) : 0;
        return v > 0 ? (uint32_t)v : 4u; // HIP's default
    }();
    return queues;
}
uint32_t wf_stream_lanes(const rt_renderer* r) {
    if (r->sched.stream_lanes) return std::min(r->sched.stream_lanes, 8u); // explicit: the caller's business
    if (wf_one_launch(r)) return 1u;
    uint32_t want = r->sched.fused_bounce || r->sched.finish_depth >= r->max_depth ? 3u : 2u;
    while (want > 2u && want + 2u > r->hw_queues) want--;
    return want;
}
// ... and where the tile has no more pixels than the chip has resident lanes (one generation of pixel chains: DESIGN.md §7) the queue
// hands every wave pixels spread evenly over the tile
bool wf_spread(const rt_renderer* r) {
    return wf_one_launch(r) && wf_stream_lanes(r) == 1 && (uint64_t)r->n_local <= (uint64_t)r->n_cus * 4u * kMegaWaves * 64u;
}
bool wf_uses_requeue(const rt_renderer* r) { return r->sched.requeue != 0 && !wf_one_launch(r); }
// Longest chains first (rt_kernels.h, k_wf_order_tiles): in the one-launch schedule, for tiles of more than one pixel generation, sample 0
// of every pixel is traced in a launch of its own and the launch for the other samples takes the 8x8 blocks in the order of what their
// first samples cost. rt_schedule::cost_order: -1 automatic (below), 0 off, 1 whenever the tile consists of whole 8x8 blocks.
// (automatic: from 32 spp on: the extra launch and the sort cost ~1 ms at 4K, 40 % of the cube's 3 ms frame at 16 spp and nothing next to 64 samples)
// ... and up to four pixel generations: with more, the last generation is a small part of the frame and the order changes nothing
// (atrium 1080p, 5.3 generations: +0.3 ms for the extra launch)
bool wf_cost_order(const rt_renderer* r) {
    if (r->sched.cost_order == 0 || !wf_one_launch(r) || wf_stream_lanes(r) != 1) return false;
    if (r->sched.cost_order > 0) return true;
    return !wf_spread(r) && r->spp >= 32 && (uint64_t)r->n_local <= 4ull * r->n_cus * 4u * kMegaWaves * 64u;
}

// samples one k_wf_finish launch renders per pixel (1 = a launch per sample). More than one only when k_wf_finish takes over from the
// camera ray on; with the dynamic queue bounded by the entry's 8-bit sample field and the queue's size (<= 1 GiB per stream lane)
uint32_t wf_chunk(const rt_renderer* r, uint32_t n_local) {
    if (r->sched.finish_depth != 0 || r->max_depth == 0 || n_local == 0) return 1u;
    if (wf_one_launch(r)) return r->spp;
    uint64_t c = std::min<uint32_t>(std::max(1u, r->sched.samples_per_launch), r->spp);
    if (wf_uses_requeue(r)) {
        c = std::min<uint64_t>(c, 256u);
        c = std::min<uint64_t>(c, std::max<uint64_t>(1, ((1ull << 30) / 8) / n_local + 1));
        if (n_local > (1u << 24)) c = 1; // the entry's pixel field has 24 bits
    }
    return (uint32_t)std::max<uint64_t>(1, c);
}

// The stream lanes' HIP streams: a pool of THREE per device, created by the first renderer on the device and shared by every renderer
// there; lane k runs on pool stream k (a lane beyond the pool — an explicit stream_lanes > 3 — gets a stream of its renderer's own).
// Rounds 1-3 created a stream per renderer and lane. Which hardware queue a stream lands on, and how well queues run side by side, turned
// out to depend on every stream the process had created before (measured in round 4 with bench.py's four renderers one after the other and a
// torch process group up; the split per-bounce pipeline, three lanes, ms per 64-spp frame):
//     a stream per renderer and lane, default priority             313   (191 when that renderer was the process's first)
//     ... at the high stream priority (a queue pool of their own)   221
//     this pool, streams created once, queues made at first use     191   in every order of the renderers
//     the same pool with a FOURTH stream that had been used once    222   (an idle fourth queue at that priority costs 16 %)
// So: few streams, created once, at a priority level the host framework's streams (torch's, RCCL's) do not use. HIP keeps a hardware-queue
// pool per priority level and makes a stream's queue at its first submission. (Two renderers of one device with per-bounce frames in
// flight at the same time share the lanes' streams: lane k of one frame queues behind lane k of the other. The one-launch schedule does
// not use lane streams.)
constexpr uint32_t kLanePool = 3;
hipError_t lane_stream_of(int device, uint32_t k, hipStream_t* out, bool* owned) {
    static std::mutex mu;
    static std::map<int, std::vector<hipStream_t>> pools;
    int prio_lo = 0, prio_hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (e != hipSuccess) return e;
    if (k >= kLanePool) { // beyond the pool (an explicit stream_lanes > 3): a stream of the renderer's own
        *owned = true;
        return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio_hi);
    }
    std::lock_guard<std::mutex> lock(mu);
    std::vector<hipStream_t>& pool = pools[device];
    if (pool.empty()) {
        std::vector<hipStream_t> fresh(kLanePool, nullptr);
        for (uint32_t i = 0; i < kLanePool; ++i) {
            e = hipStreamCreateWithPriority(&fresh[i], hipStreamNonBlocking, prio_hi);
            if (e != hipSuccess) {
                for (uint32_t j = 0; j < i; ++j) (void)hipStreamDestroy(fresh[j]);
                return e;
            }
        }
        pool = fresh; // (lives as long as the process: a handful of idle streams)
    }
    *out = pool[k], *owned = false;
    return hipSuccess;
}

int alloc_tile_buffers_impl(rt_renderer* r) {
    free_tile_buffers(r);
    // test hook, DEVELOPER builds only (rt_knobs.h): RT_INJECT_ALLOC_FAILURE=k makes the k-th device allocation of this call fail as if the device
    // were out of memory (tests/test_gpu_parity.py: a failed re-allocation must leave a renderer that says so, not one that launches on null queues)
    int inject = 0, n_alloc = 0;
    if (const char* e = dev_knob("RT_INJECT_ALLOC_FAILURE")) inject = std::atoi(e);
    auto dalloc = [&](void** p, size_t bytes) { return inject && ++n_alloc == inject ? hipErrorOutOfMemory : hipMalloc(p, bytes); };
    r->tile.local_rows = rows_of_tile(r->tile, r->height);
    r->n_local = (uint32_t)r->tile.local_rows * (uint32_t)r->width;
    const size_t n = std::max<size_t>(r->n_local, 1);
    const char* ks = std::getenv("RT_KERNEL_STATS");
    if (ks && ks[0] == '1') HIPCHK(dalloc((void**)&r->d_stats, 24 * sizeof(unsigned long long)));
    // sized for the LONGEST tile of the split (rank 0's): ncclGather sends the same count from every rank
    TileDev t0 = r->tile;
    t0.rank = 0;
    const size_t n_pad = std::max<size_t>((size_t)rows_of_tile(t0, r->height) * (size_t)r->width, n);
    HIPCHK(dalloc((void**)&r->d_f32, n_pad * 16));
    HIPCHK(dalloc((void**)&r->d_u8, n_pad * 4));
    // the padding rows are gathered too: defined bytes. On the renderer's own stream: a hipMemset on the null stream is not ordered
    // against this non-blocking stream and could land on top of the first frame.
    HIPCHK(hipMemsetAsync(r->d_f32, 0, n_pad * 16, r->stream));
    HIPCHK(hipMemsetAsync(r->d_u8, 0, n_pad * 4, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));
    if (r->kind == RT_RENDERER_MEGAKERNEL) {
        HIPCHK(dalloc((void**)&r->d_ray_counter, 8 * sizeof(unsigned long long))); // [0] ray count, [1] the frame's pixel cursor (low word)
        if (r->sched.pixel_slices != 0 && r->sched.pixel_slices != 1 && (uint64_t)n * 32u < (1ull << 31)) { // (the kernel addresses the state with 32-bit offsets)
            HIPCHK(dalloc((void**)&r->d_slice_state, n * 32));
            HIPCHK(hipMemsetAsync(r->d_slice_state, 0, n * 32, r->stream));
            HIPCHK(hipStreamSynchronize(r->stream));
            r->slice_tag = 0;
        }
    } else {
        const uint32_t K = wf_stream_lanes(r);
        const bool per_bounce = r->sched.finish_depth > 0 && r->max_depth > 0; // EXTEND + SHADE launch pairs: the second queue and the hit records
        r->lanes.resize(K);
        for (uint32_t k = 0; k < K; ++k) {
            auto& L = r->lanes[k];
            // sub-tile k of K inside tile (rank, world) == tile (k*world + rank) of (K*world): see DESIGN.md §7
            L.tile = TileDev{k * r->tile.world + r->tile.rank, K * r->tile.world, r->tile.strip_rows, 0};
            L.tile.local_rows = rows_of_tile(L.tile, r->height);
            L.n_local = (uint32_t)L.tile.local_rows * (uint32_t)r->width;
            const size_t m = std::max<size_t>(L.n_local, 1);
            // SHOOT launches reserve the next queue in blocks of kOutBlock slots per workgroup: up to one partly used block per workgroup
            const size_t mq = m + (per_bounce && r->sched.fused_bounce ? (size_t)kOutBlock * ((size_t)r->n_cus * 4u * kMegaWaves / (kMegaBlock / 64) + 1u) : 0u);
            for (int qi = 0; qi < (per_bounce ? 2 : 1); ++qi) { // k_wf_generate -> k_wf_finish needs one queue and no hit records
                auto& q = L.q[qi];
                HIPCHK(dalloc((void**)&q.org_id, mq * 16));
                HIPCHK(dalloc((void**)&q.a, mq * 8));
                HIPCHK(dalloc((void**)&q.b, mq * 8));
                HIPCHK(dalloc((void**)&q.c, mq * 2 + 2));
            }
            HIPCHK(dalloc((void**)&L.d_rng, m * 4));
            HIPCHK(dalloc((void**)&L.d_accum, m * 16));
            if (per_bounce && !r->sched.fused_bounce) HIPCHK(dalloc((void**)&L.d_hits, m * 16));
            const size_t nc = (size_t)r->spp * ((size_t)r->max_depth + 1);
            // spp * (max_depth + 1) queue lengths, spp * max_depth chunk cursors, spp "extra ray" counters of k_wf_finish
            HIPCHK(dalloc((void**)&L.d_counts, (nc + (size_t)r->spp * r->max_depth + (size_t)r->spp + 1) * 4));
            L.h_counts.resize(nc);
            L.h_extra.resize((size_t)r->spp + 1); // + the dynamic queue's error word
            HIPCHK(dalloc((void**)&L.d_extra, L.h_extra.size() * 8));
            const uint32_t chunk = wf_chunk(r, L.n_local);
            L.n_tiles = 0;
            if (wf_cost_order(r) && r->width % 8 == 0 && L.tile.local_rows % 8 == 0 && L.n_local) {
                L.n_tiles = L.n_local / 64u;
                HIPCHK(dalloc((void**)&L.d_cost, L.n_local));
                HIPCHK(dalloc((void**)&L.d_tile_key, (size_t)L.n_tiles * 4));
                HIPCHK(dalloc((void**)&L.d_tile_order, (size_t)L.n_tiles * 4));
            }
            if (chunk > 1 && wf_uses_requeue(r)) { // every pixel is appended once per sample of a launch but its first
                L.rq_capacity = L.n_local * (chunk - 1u);
                HIPCHK(dalloc((void**)&L.d_rq_entries, (size_t)L.rq_capacity * 8));
                HIPCHK(dalloc((void**)&L.d_rq_ctl, 80 * 4));
            }
            // (a frame captured into a hipGraph forks into its lanes' streams and puts them into capture mode: such a renderer gets lane
            // streams of its OWN — on the device's shared pool another renderer's launches would be captured into this graph, or fail)
            HIPCHK(lane_stream_of(r->scene->device, r->sched.hip_graph && K > 1 ? kLanePool + k : k, &L.stream, &L.stream_owned));
            HIPCHK(hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
        }
    }
    return RT_OK;
}

// Re-allocation is all or nothing: whatever fails (a device allocation, a stream), nothing half-built survives — the renderer is marked
// unusable and every frame entry point refuses it until a later rt_renderer_set_tile / rt_renderer_set_schedule succeeds.
int alloc_tile_buffers(rt_renderer* r) {
    const int rc = alloc_tile_buffers_impl(r);
    r->unusable = rc != RT_OK;
    if (rc != RT_OK) {
        const std::string msg = g_err;
        free_tile_buffers(r);
        r->n_local = 0, r->tile.local_rows = 0;
        g_err = msg;
    }
    return rc;
}

// Pixel slices of a megakernel frame (rt_kernels.h: SliceDev). G = pixel slots per resident lane, the frame's "generations". Slice j + 1 of a
// pixel is handed out one sweep of the cursor after slice j, and a sweep over slices of length f takes G x f: the next slice is ready as long
// as slices shrink by less than G from one to the next — by G / 1.5 here, at most 4, for the spread of the pixels' costs. The frame drains
// over its LAST slice (about 1/16 of the samples), instead of over a whole pixel.
//   G <= 1.25 : one slice — nothing is handed out after the first generation, a slice would only wait for the one before it
//   forced    : rt_schedule::pixel_slices >= 2 slices of that geometry whatever G is (tests, sweeps)
SliceDev mega_slices(const rt_renderer* r, uint32_t grid, uint32_t n_slots, uint32_t tag0) {
    SliceDev sl{};
    sl.state = r->d_slice_state, sl.n_slices = 1, sl.tag0 = tag0;
    for (uint32_t k = 0; k < kMaxSlices; ++k) sl.bound[k] = r->spp;
    const int32_t want = r->sched.pixel_slices;
    const double G = (double)n_slots / ((double)grid * kMegaBlock);
    if (!r->d_slice_state || want == 0 || want == 1 || r->spp < 2 || (want < 0 && G <= 1.25)) return sl;
    // a slice ends at a multiple of 2^shift samples: the kernel finds the ends in a 64-bit mask, bit k = "a slice ends before sample k << shift"
    uint32_t shift = 0;
    while (((r->spp - 1u) >> shift) >= 64u) shift++;
    const uint32_t unit = 1u << shift, units = (r->spp + unit - 1u) / unit; // the last unit may be a short one
    if (units < 2) return sl;
    const double ratio = std::min(4.0, std::max(1.0, G / 1.5));
    // (measured, atrium 1080p 64 spp, profiles/r05_slice_sweep.txt: slices from 48, 60 on 111.7 ms, from 46, 59, 63 on 112.7, from 56 on 114.9, unsliced
    // 119.9; half the frame: from 32, 48, 56, 60 on 57.4 ms, unsliced 67.0; a quarter: eight equal slices 32.7 ms, unsliced 42.5 — a last slice
    // of 1/16 of the samples is short enough, shorter ones only add hand-overs)
    const double last = std::max(1.0, units / 16.0);
    uint32_t n = want >= 2 ? (uint32_t)want : (ratio > 1.05 ? 1u + (uint32_t)std::floor(std::log(units / last) / std::log(ratio) + 1e-9) : kMaxSlices);
    n = std::max(n, 2u);
    n = std::min(std::min(n, kMaxSlices), units);
    double wsum = 0.0;
    for (uint32_t i = 0; i < n; ++i) wsum += std::pow(ratio, (double)(n - 1 - i));
    // sizes (in units) from the last slice up, each at least one unit; the first slice takes what is left
    std::vector<uint32_t> size(n, 1u);
    uint32_t used = 0;
    for (uint32_t i = n; i-- > 1;) {
        size[i] = std::max<uint32_t>(1u, (uint32_t)std::lround(units * std::pow(ratio, (double)(n - 1 - i)) / wsum));
        size[i] = std::min(size[i], units - used - i); // leave one unit for every earlier slice
        used += size[i];
    }
    size[0] = units - used;
    uint32_t b = 0;
    for (uint32_t i = 0; i + 1 < n; ++i) {
        b += size[i];
        sl.bound[i] = b * unit; // (< spp: at least one unit follows)
        sl.cuts |= 1ull << b;
    }
    sl.shift = shift;
    sl.n_slices = n;
    if (const char* e = dev_knob("RT_MEGA_SLICE_BOUNDS")) { // sweeps: the samples slices 1, 2, ... start with, ascending, e.g. "48,60,63" (shift 0 only)
        sl.cuts = 0, sl.n_slices = 1;
        for (uint32_t k = 0; k < kMaxSlices; ++k) sl.bound[k] = r->spp;
        uint32_t prev = 0;
        for (const char* p = e; *p && sl.n_slices < kMaxSlices && shift == 0;) {
            const uint32_t v = (uint32_t)std::strtoul(p, const_cast<char**>(&p), 10);
            if (*p == ',') ++p;
            if (v <= prev || v >= r->spp) break;
            sl.bound[sl.n_slices - 1] = v, sl.cuts |= 1ull << v, sl.n_slices++, prev = v;
        }
    }
    return sl;
}

CameraDev to_dev(const rt_camera& c) {
    CameraDev d;
    std::memcpy(d.center, c.center, 12), std::memcpy(d.pixel00, c.pixel00, 12);
    std::memcpy(d.du, c.delta_u, 12), std::memcpy(d.dv, c.delta_v, 12);
    d.width = c.width, d.height = c.height;
    return d;
}

hipEvent_t pool_event(std::vector<hipEvent_t>& pool, size_t i) {
    while (pool.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        pool.push_back(e);
    }
    return pool[i];
}

// The wavefront frame: every launch of the lanes' chains, forked from and joined back into `st`. Issued directly
// or recorded by a stream capture (enqueue_frame).
// Under stream capture (rt_schedule::hip_graph) the frame's small fills and its camera upload are KERNEL nodes: as memset / memcpy nodes of a
// replayed graph they were not reliable on ROCm 7.2 — from the third replay on, the queue-length words of a renderer whose buffers had been
// re-allocated before the capture came back holding foreign 16-byte patterns (the frame still rendered, its ray count did not:
// profiles/r05_graph_nodes.txt). A kernel reads the pinned host copy of the camera through its device address.
__global__ void __launch_bounds__(256) k_fill_words(uint32_t* __restrict__ p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(64) k_copy_words(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t n) {
    if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
}
int fill_bytes(void* p, int byte, size_t bytes, hipStream_t st, bool capturing) { // bytes: a multiple of 4
    if (!capturing) {
        HIPCHK(hipMemsetAsync(p, byte, bytes, st));
        return RT_OK;
    }
    const uint32_t n = (uint32_t)(bytes / 4), b = (uint32_t)(byte & 0xFF);
    if (n) hipLaunchKernelGGL(k_fill_words, dim3((n + 255u) / 256u), dim3(256), 0, st, (uint32_t*)p, n, b * 0x01010101u);
    return RT_OK;
}

int enqueue_wavefront(rt_renderer* r, float* d_f32, uint8_t* d_u8, hipStream_t st, LaunchCount* launches_out, size_t* n_hot_events, bool capturing = false) {
    const SceneDev& S = r->scene->dev;
    LaunchCount launches;
    if (capturing) {
        static_assert(sizeof(CameraDev) % 4 == 0 && sizeof(CameraDev) / 4 <= 64, "k_copy_words copies the camera with one wave");
        void* h_cam_dev = nullptr;
        HIPCHK(hipHostGetDevicePointer(&h_cam_dev, r->h_cam, 0));
        hipLaunchKernelGGL(k_copy_words, dim3(1), dim3(64), 0, st, (uint32_t*)r->d_cam, (const uint32_t*)h_cam_dev, (uint32_t)(sizeof(CameraDev) / 4));
    } else {
        HIPCHK(hipMemcpyAsync(r->d_cam, r->h_cam, sizeof(CameraDev), hipMemcpyHostToDevice, st));
    }
    const uint32_t D = r->max_depth;
    if (r->d_stats) { int rc_ = fill_bytes(r->d_stats, 0, 24 * sizeof(unsigned long long), st, capturing); if (rc_ != RT_OK) return rc_; }
    // fork: every lane's stream waits for what was enqueued on `st` so far, then runs its own chain
    if (r->lanes.size() > 1) HIPCHK(hipEventRecord(r->ev_fork, st));
    size_t ne_total = 0;
    const uint32_t K = (uint32_t)r->lanes.size();
    // one stream lane (the default schedule): its launches go straight onto `st` — no fork / join events, no second stream to wake up
    auto lane_stream = [&](rt_renderer::WfLane& L) { return K == 1 ? st : L.stream; };
    constexpr uint32_t ext_wpb = kExtendBlock / 64;
    std::vector<uint32_t> grid(K), ext_grid(K), buf(K, 0u);
    std::vector<size_t> ne(K, 0);
    for (uint32_t k = 0; k < K; ++k) {
        auto& L = r->lanes[k];
        L.n_hot = 0;
        L.spans.clear();
        if (K > 1) HIPCHK(hipStreamWaitEvent(lane_stream(L), r->ev_fork, 0));
        if (L.n_local == 0) continue;
        grid[k] = (L.n_local + 255u) / 256u;
        // persistent EXTEND waves: fill every SIMD at the kernel's occupancy, never more than there are chunks
        const uint32_t ext_waves = std::max<uint32_t>(1u, std::min<uint32_t>((L.n_local + kExtendChunk - 1u) / kExtendChunk,
                                                                            (uint32_t)r->n_cus * 4u * (r->wf_extend_occ ? r->wf_extend_occ : (K >= 2 ? kExtendWaves / 2 : kExtendWaves))));
        ext_grid[k] = (ext_waves + ext_wpb - 1u) / ext_wpb;
        if (int rc_ = fill_bytes(L.d_counts, 0, (L.h_counts.size() + (size_t)r->spp * D + (size_t)r->spp + 1) * 4, lane_stream(L), capturing)) return rc_;
        if (int rc_ = fill_bytes(L.d_extra, 0, L.h_extra.size() * 8, lane_stream(L), capturing)) return rc_;
        hipLaunchKernelGGL(k_wf_init, dim3(grid[k]), dim3(256), 0, lane_stream(L), r->d_cam, L.tile, r->seed_mode, L.n_local, L.d_rng, L.d_accum);
        launches.add(RT_K_WF_INIT);
    }
    // the lanes' chains are enqueued round-robin, bounce by bounce, so their kernels interleave on the device
    const uint32_t D0 = std::min(D, r->sched.finish_depth); // bounces done launch by launch; the rest of every path in k_wf_finish
    // samples one k_wf_finish launch renders per pixel: when it takes over from the camera ray on (D0 == 0) a lane whose path has ended
    // generates the pixel's NEXT sample itself and goes on, for `chunk` samples, before it hands the RNG word back to memory
    uint32_t chunk = r->spp;
    for (uint32_t k = 0; k < K; ++k)
        if (r->lanes[k].n_local) chunk = std::min(chunk, wf_chunk(r, r->lanes[k].n_local));
    // cost ordering (one-launch schedule, one stream lane): sample 0 in a launch of its own, then all the others, most expensive blocks first
    const bool lpt = K == 1 && r->lanes[0].n_tiles != 0 && chunk == r->spp && D0 == 0 && D > 0;
    const bool shoot_frame = r->sched.fused_bounce && D0 > 0; // queues may hold holes: every launch counts its rays on the device
    for (uint32_t s = 0; s < r->spp;) {
        const uint32_t n_samples = lpt ? (s == 0 ? 1u : r->spp - 1u) : std::min(chunk, r->spp - s);
        const bool ordered = lpt && s > 0;
        for (uint32_t k = 0; k < K; ++k) {
            auto& L = r->lanes[k];
            if (L.n_local == 0) continue;
            buf[k] = 0;
            if (ordered) {
                hipLaunchKernelGGL(k_wf_tile_cost, dim3(L.n_tiles), dim3(64), 0, lane_stream(L), L.d_cost, (uint32_t)r->width, L.n_tiles, L.d_tile_key);
                hipLaunchKernelGGL(k_wf_order_tiles, dim3(1), dim3(1024), 0, lane_stream(L), L.d_tile_key, L.n_tiles, L.d_tile_order);
                launches.add(RT_K_WF_TILE_ORDER, 2);
            }
            hipLaunchKernelGGL(k_wf_generate, dim3(grid[k]), dim3(256), 0, lane_stream(L), r->d_cam, L.tile, L.n_local, L.d_rng, L.q[0],
                               L.d_counts + (size_t)s * (D + 1), wf_spread(r) ? 1u : 0u, ordered ? L.d_tile_order : (const uint32_t*)nullptr);
            launches.add(RT_K_WF_GENERATE);
        }
        for (uint32_t d = 0; d < D0; ++d) {
            for (uint32_t k = 0; k < K; ++k) {
                auto& L = r->lanes[k];
                if (L.n_local == 0) continue;
                hipStream_t ls = lane_stream(L);
                uint32_t* counts = L.d_counts + (size_t)s * (D + 1);
                uint32_t* cursors = L.d_counts + L.h_counts.size();
                if (r->sched.fused_bounce) { // SHOOT: intersect + shade + compact in ONE launch per bounce (k_wf_finish<.., LIMIT> with a limit of one bounce)
                    constexpr uint32_t fin_wpb = kMegaBlock / 64;
                    const uint32_t fin_cap = (uint32_t)r->n_cus * 4u * (r->wf_finish_occ ? r->wf_finish_occ : std::max<uint32_t>(1u, kMegaWaves / K));
                    const uint32_t chunk_slots = r->wf_shoot_chunk ? r->wf_shoot_chunk : r->wf_shoot_tail ? 256u : (L.n_local < 2u * 64u * fin_cap ? 64u : kExtendChunk);
                    const uint32_t first_slots = r->wf_shoot_tail ? 64u : chunk_slots; // (guided: the kernel sizes the static part from the queue's length; 64 = at least)
                    const uint32_t fin_waves = std::max<uint32_t>(1u, std::min<uint32_t>((L.n_local + first_slots - 1u) / first_slots, fin_cap));
                    const uint32_t fin_grid = (fin_waves + fin_wpb - 1u) / fin_wpb;
                    const uint32_t chunk_arg = chunk_slots | (r->wf_shoot_tail ? 0x80000000u | (r->wf_shoot_static_pct << 16) : 0u);
                    hipEvent_t f0 = nullptr, f1 = nullptr;
                    if (r->profile_kernels) {
                        f0 = pool_event(L.ev_pool, ne[k]), f1 = pool_event(L.ev_pool, ne[k] + 1);
                        if (!f0 || !f1) return fail(RT_ERR_HIP, "hipEventCreate failed");
                        HIPCHK(hipEventRecord(f0, ls));
                    }
                    const FinishRequeue no_fq{nullptr, nullptr, reinterpret_cast<uint32_t*>(L.d_extra + r->spp), 0u}; // only the error word
#define RT_SHOOT_LAUNCH(ST)                                                                                                                  \
    hipLaunchKernelGGL((k_wf_finish<ST, false, true>), dim3(fin_grid), dim3(kMegaBlock), 0, ls, S, r->d_cam, L.tile, L.q[buf[k]], counts + d,    \
                       cursors + (size_t)s * D + d, L.d_rng, L.d_accum, L.d_extra + s, d, D, 1u, r->rr_start,                                 \
                       ST ? r->d_stats : (unsigned long long*)nullptr, no_fq, chunk_arg, (uint8_t*)nullptr, L.q[buf[k] ^ 1], counts + d + 1, 1u, 1u)
                    if (r->d_stats) RT_SHOOT_LAUNCH(true);
                    else RT_SHOOT_LAUNCH(false);
#undef RT_SHOOT_LAUNCH
                    if (r->profile_kernels) {
                        HIPCHK(hipEventRecord(f1, ls));
                        L.spans.push_back({(uint32_t)ne[k], (uint32_t)ne[k] + 1, RT_K_WF_SHOOT});
                        ne[k] += 2;
                    }
                    launches.add(RT_K_WF_SHOOT);
                    buf[k] ^= 1;
                    continue;
                }
                hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr; // profiling: e0 EXTEND e1 SHADE e2
                if (r->profile_kernels) {
                    e0 = pool_event(L.ev_pool, ne[k]), e1 = pool_event(L.ev_pool, ne[k] + 1), e2 = pool_event(L.ev_pool, ne[k] + 2);
                    if (!e0 || !e1 || !e2) return fail(RT_ERR_HIP, "hipEventCreate failed");
                    HIPCHK(hipEventRecord(e0, ls));
                }
                if (r->d_stats)
                    hipLaunchKernelGGL(k_wf_extend<true>, dim3(ext_grid[k]), dim3(kExtendBlock), 0, ls, S, L.q[buf[k]], L.d_hits,
                                       counts + d, cursors + (size_t)s * D + d, r->d_stats);
                else
                    hipLaunchKernelGGL(k_wf_extend<false>, dim3(ext_grid[k]), dim3(kExtendBlock), 0, ls, S, L.q[buf[k]], L.d_hits,
                                       counts + d, cursors + (size_t)s * D + d, (unsigned long long*)nullptr);
                if (r->profile_kernels) HIPCHK(hipEventRecord(e1, ls));
                {
                    const int last = d + 1 == D ? 1 : 0, rr = r->rr_start && d + 1 >= r->rr_start && d + 1 < D ? 1 : 0;
#define RT_SHADE_LAUNCH(RE, MS)                                                                                                         \
    hipLaunchKernelGGL((k_wf_shade<RE, MS>), dim3(grid[k]), dim3(256), 0, ls, S, L.q[buf[k]], L.q[buf[k] ^ 1], L.d_hits, L.d_rng, L.d_accum, \
                       counts + d, counts + d + 1, last, rr)
                    if (r->sched.reorder && r->sched.matsort) RT_SHADE_LAUNCH(true, true);
                    else if (r->sched.reorder) RT_SHADE_LAUNCH(true, false);
                    else if (r->sched.matsort) RT_SHADE_LAUNCH(false, true);
                    else RT_SHADE_LAUNCH(false, false);
#undef RT_SHADE_LAUNCH
                }
                if (r->profile_kernels) {
                    HIPCHK(hipEventRecord(e2, ls));
                    L.spans.push_back({(uint32_t)ne[k], (uint32_t)ne[k] + 1, RT_K_WF_EXTEND});
                    L.spans.push_back({(uint32_t)ne[k] + 1, (uint32_t)ne[k] + 2, RT_K_WF_SHADE});
                    ne[k] += 3;
                }
                launches.add(RT_K_WF_EXTEND), launches.add(RT_K_WF_SHADE);
                if (r->sched.reorder) launches.also(RT_K_WF_SHADE_REORDER);
                if (r->sched.matsort) launches.also(RT_K_WF_SHADE_MATSORT);
                buf[k] ^= 1;
            }
        }
        if (D0 < D) {
            constexpr uint32_t fin_wpb = kMegaBlock / 64;
            for (uint32_t k = 0; k < K; ++k) {
                auto& L = r->lanes[k];
                if (L.n_local == 0) continue;
                hipStream_t ls = lane_stream(L);
                uint32_t* counts = L.d_counts + (size_t)s * (D + 1);
                uint32_t* cursors = L.d_counts + L.h_counts.size();
                // persistent grid: wf_finish_occ waves per SIMD for THIS launch. The lanes' launches overlap on the device, so each asks for
                // its share of the SIMDs' wave slots rather than all of them: fewer, longer-lived waves follow more paths per lane
                // (less idling at the end of each wave's last paths) and the streams' kernels are resident side by side.
                const uint32_t fin_cap = (uint32_t)r->n_cus * 4u * (r->wf_finish_occ ? r->wf_finish_occ : std::max<uint32_t>(1u, kMegaWaves / K));
                // queue slots a wave takes at a time: 128 as EXTEND does — but 64 (one per lane) when that is what it takes to give every
                // resident wave something to do (a 1/8 tile has fewer pixels than the chip has lanes)
                const uint32_t chunk_slots = (L.n_local < 2u * 64u * fin_cap || (n_samples > 1 && !L.d_rq_entries)) ? 64u : kExtendChunk;
                const uint32_t fin_waves = std::max<uint32_t>(1u, std::min<uint32_t>((L.n_local + chunk_slots - 1u) / chunk_slots, fin_cap));
                const uint32_t fin_grid = (fin_waves + fin_wpb - 1u) / fin_wpb;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (r->profile_kernels) { // the dominant kernel of this schedule
                    e0 = pool_event(L.ev_pool, ne[k]), e1 = pool_event(L.ev_pool, ne[k] + 1);
                    if (!e0 || !e1) return fail(RT_ERR_HIP, "hipEventCreate failed");
                    HIPCHK(hipEventRecord(e0, ls));
                }
                // rays that are not a queue's length (every bounce beyond D0, every ray of the samples generated in-lane) are counted here
                unsigned long long* extra = L.d_extra + s;
                const bool req = n_samples > 1 && L.d_rq_entries != nullptr;
                FinishRequeue fq{L.d_rq_entries, L.d_rq_ctl, reinterpret_cast<uint32_t*>(L.d_extra + r->spp), L.rq_capacity};
                if (req) { // the launch's dynamic queue: all entries invalid, counters zero
                    if (int rc_ = fill_bytes(L.d_rq_entries, 0xFF, (size_t)L.n_local * (n_samples - 1u) * 8, ls, capturing)) return rc_;
                    if (int rc_ = fill_bytes(L.d_rq_ctl, 0, 80 * 4, ls, capturing)) return rc_;
                }
#define RT_FINISH_LAUNCH(ST, RQ)                                                                                                              \
    hipLaunchKernelGGL((k_wf_finish<ST, RQ>), dim3(fin_grid), dim3(kMegaBlock), 0, ls, S, r->d_cam, L.tile, L.q[buf[k]], counts + D0,        \
                       cursors + (size_t)s * D + D0, L.d_rng, L.d_accum, extra, D0, D, n_samples, r->rr_start,                                 \
                       ST ? r->d_stats : (unsigned long long*)nullptr, fq, chunk_slots, lpt && s == 0 ? L.d_cost : (uint8_t*)nullptr, QueueDev{}, (uint32_t*)nullptr, 0u,     \
                       shoot_frame ? 1u : 0u)
                if (r->d_stats && req) RT_FINISH_LAUNCH(true, true);
                else if (r->d_stats) RT_FINISH_LAUNCH(true, false);
                else if (req) RT_FINISH_LAUNCH(false, true);
                else RT_FINISH_LAUNCH(false, false);
#undef RT_FINISH_LAUNCH
                if (r->profile_kernels) {
                    HIPCHK(hipEventRecord(e1, ls));
                    L.spans.push_back({(uint32_t)ne[k], (uint32_t)ne[k] + 1, RT_K_WF_FINISH});
                    ne[k] += 2;
                }
                launches.add(RT_K_WF_FINISH);
                if (req) launches.also(RT_K_WF_FINISH_REQUEUE);
            }
        }
        s += n_samples;
    }
    HIPCHK(hipGetLastError());
    for (uint32_t k = 0; k < K; ++k) {
        auto& L = r->lanes[k];
        if (L.n_local) {
            hipLaunchKernelGGL(k_wf_resolve, dim3(grid[k]), dim3(256), 0, lane_stream(L), L.n_local, r->spp, (uint32_t)r->width,
                               r->tile.strip_rows, K, k, L.d_accum, d_f32, d_u8);
            launches.add(RT_K_WF_RESOLVE);
        }
        L.n_hot = ne[k];
        ne_total += ne[k];
        // join: `st` continues only after this lane's chain is complete
        if (K > 1) {
            HIPCHK(hipEventRecord(L.ev_done, lane_stream(L)));
            HIPCHK(hipStreamWaitEvent(st, L.ev_done, 0));
        }
    }
    HIPCHK(hipGetLastError());
    *n_hot_events = ne_total;
    *launches_out = launches;
    return RT_OK;
}

// Enqueues one frame on `st`. Device outputs may be null.
int enqueue_frame(rt_renderer* r, const rt_camera* cam, float* d_f32, uint8_t* d_u8, hipStream_t st, rt_stats* stats,
                  size_t* n_hot_events) {
    const CameraDev c = to_dev(*cam);
    const SceneDev& S = r->scene->dev;
    LaunchCount launches;
    *n_hot_events = 0;
    if (r->n_local == 0) {
        if (stats) stats->launches = 0;
        return RT_OK;
    }
    if (r->kind == RT_RENDERER_MEGAKERNEL) {
        HIPCHK(hipMemsetAsync(r->d_ray_counter, 0, 8 * sizeof(unsigned long long), st)); // ray count + pixel cursor
        const uint32_t tiles_x = (uint32_t)(r->width + 7) / 8u, tiles_y = (uint32_t)(r->tile.local_rows + 7) / 8u;
        // persistent waves: enough to fill every SIMD at the kernel's occupancy, never more than there are tiles
        constexpr uint32_t mega_wpb = kMegaBlock / 64;
        // (in the chain regime — fewer tiles than wave slots — 64 pixels on 4,050 waves beat 42 pixels on all 6,144: 29.5 against 33.7 ms)
        const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>((tiles_x * tiles_y + mega_wpb - 1u) / mega_wpb,
                                                                        (uint32_t)r->n_cus * 4u * r->mega_occ / mega_wpb));
        const uint32_t pad = r->mega_lds_pad;
        if (pad) { // static + dynamic LDS may exceed the 64 KiB a launch gets without asking
            (void)hipFuncSetAttribute((const void*)k_megakernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            (void)hipFuncSetAttribute((const void*)k_megakernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            (void)hipFuncSetAttribute((const void*)k_megakernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            (void)hipFuncSetAttribute((const void*)k_megakernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
        }
        hipEvent_t e0 = pool_event(r->ev_pool, 0), e1 = pool_event(r->ev_pool, 1);
        if (!e0 || !e1) return fail(RT_ERR_HIP, "hipEventCreate failed");
        // tags of this frame: tag0 + (first sample of the slice the state is for), tag0 > every tag of the frames before
        if (r->d_slice_state && r->slice_tag > 0xFFFFFFFFu - 2u * (r->spp + 1u)) { // about to wrap: start over from a cleared buffer
            HIPCHK(hipMemsetAsync(r->d_slice_state, 0, (size_t)r->n_local * 32, st));
            r->slice_tag = 0;
        }
        r->slice_tag += r->spp + 1u;
        const SliceDev sl = mega_slices(r, grid, tiles_x * tiles_y * 64u, r->slice_tag);
        if (stats) stats->pixel_slices = sl.n_slices;
        // (h_frame is rewritten per frame: the previous frame of this renderer has been collected, frame_begin refuses a second one in flight)
        r->h_frame->cam = c, r->h_frame->sl = sl;
        HIPCHK(hipMemcpyAsync(r->d_frame, r->h_frame, sizeof(MegaFrame), hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(e0, st));
        if (r->max_depth == 0) {
            hipLaunchKernelGGL(k_fill_black, dim3((r->n_local + 255u) / 256u), dim3(256), 0, st, r->n_local, d_f32, d_u8);
        } else if (r->d_stats) {
            HIPCHK(hipMemsetAsync(r->d_stats, 0, 13 * sizeof(unsigned long long), st));
            HIPCHK(hipMemsetAsync(r->d_stats + 13, 0xff, 2 * sizeof(unsigned long long), st)); // the two atomicMin slots
            HIPCHK(hipMemsetAsync(r->d_stats + 15, 0, 9 * sizeof(unsigned long long), st));
#define RT_MEGA_LAUNCH(ST, SL)                                                                                                                    \
    hipLaunchKernelGGL((k_megakernel<ST, SL>), dim3(grid), dim3(kMegaBlock), pad, st, S, r->d_frame, c.width, c.height, r->tile, r->max_depth, r->spp, \
                       r->seed_mode, r->rr_start, d_f32, d_u8, r->d_ray_counter, ST ? r->d_stats : (unsigned long long*)nullptr, sl.n_slices)
            if (sl.n_slices > 1u) RT_MEGA_LAUNCH(true, true);
            else RT_MEGA_LAUNCH(true, false);
        } else {
            if (sl.n_slices > 1u) RT_MEGA_LAUNCH(false, true);
            else RT_MEGA_LAUNCH(false, false);
#undef RT_MEGA_LAUNCH
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e1, st));
        *n_hot_events = 2;
        launches.add(r->max_depth == 0 ? RT_K_FILL_BLACK : RT_K_MEGAKERNEL);
        if (stats) stats->stream_lanes = 1, stats->samples_per_launch = r->spp;
    } else {
        *r->h_cam = c;
        size_t ne_total = 0;
        if (!(r->sched.hip_graph && !r->profile_kernels && !r->d_stats)) {
            int rc = enqueue_wavefront(r, d_f32, d_u8, st, &launches, &ne_total);
            if (rc != RT_OK) return rc;
        } else {
            if (!r->graph_exec || r->graph_f32 != d_f32 || r->graph_u8 != d_u8 || r->graph_stream != st) {
                drop_graph(r);
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
                const int rc = enqueue_wavefront(r, d_f32, d_u8, st, &r->graph_launches, &ne_total, true);
                const hipError_t ce = hipStreamEndCapture(st, &g); // also ends a capture that failed half-way
                if (rc != RT_OK || ce != hipSuccess || !g) {
                    if (g) (void)hipGraphDestroy(g);
                    return rc != RT_OK ? rc : fail(RT_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
                }
                const hipError_t ie = hipGraphInstantiate(&r->graph_exec, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (ie != hipSuccess) {
                    r->graph_exec = nullptr;
                    return fail(RT_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
                }
                r->graph_f32 = d_f32, r->graph_u8 = d_u8, r->graph_stream = st;
            }
            HIPCHK(hipGraphLaunch(r->graph_exec, st));
            launches = r->graph_launches;
        }
        *n_hot_events = ne_total;
        if (stats) { // the schedule as it was resolved for this tile
            stats->stream_lanes = (uint32_t)r->lanes.size();
            stats->finish_depth = std::min(r->max_depth, r->sched.finish_depth);
            uint32_t chunk = r->spp;
            for (auto& L : r->lanes)
                if (L.n_local) chunk = std::min(chunk, wf_chunk(r, L.n_local));
            stats->samples_per_launch = chunk;
            stats->cost_ordered = launches.by[RT_K_WF_TILE_ORDER] ? 1u : 0u;
        }
    }
    if (stats) {
        stats->launches = launches.total;
        std::memcpy(stats->launches_by_kernel, launches.by, sizeof(launches.by));
    }
    return RT_OK;
}

// A frame is enqueued (frame_begin) and collected (frame_end) separately, so that a caller can keep a second frame
// of ANOTHER renderer in flight meanwhile: its persistent waves move in as this frame's last pixels drain.
int frame_begin(rt_renderer* r, const rt_camera* cam, float* d_f32, uint8_t* d_u8, hipStream_t st) {
    if (!r || !cam) return fail(RT_ERR_INVALID, "null renderer or camera");
    if (r->frame_pending) return fail(RT_ERR_INVALID, "the renderer's previous frame has not been collected (rt_render_frame_end)");
    if (r->unusable) return fail(RT_ERR_INVALID, "the renderer has no buffers: its last rt_renderer_set_tile / rt_renderer_set_schedule failed (call either again)");
    if (cam->width != r->width || cam->height != r->height) return fail(RT_ERR_INVALID, "camera size differs from the renderer's image size");
    if (!origin_in_contract_range(r->scene->hs, cam->center))
        return fail(RT_ERR_INVALID, "the camera lies more than 100 scene scales outside the scene's bounds: outside the range of the closest-hit contract (rt_intersect_batch)");
    HIPCHK(hipSetDevice(r->scene->device));
    r->pending_stats = rt_stats{};
    r->pending_stats.hw_queues = r->hw_queues;
    r->pending_hot = 0;
    r->pending_t0 = std::chrono::high_resolution_clock::now();
    if (r->tile_read_recorded && st != r->stream) HIPCHK(hipStreamWaitEvent(st, r->ev_tile_read, 0)); // an overlapped gather still reads the tile
    HIPCHK(hipEventRecord(r->ev_begin, st));
    int rc = enqueue_frame(r, cam, d_f32, d_u8, st, &r->pending_stats, &r->pending_hot);
    if (rc != RT_OK) {
        // part of the frame may already be queued (on `st` and on the wavefront lanes' streams): nothing may still be running on the
        // tile buffers when the caller reacts to the error (set_tile / destroy free them), so drain before reporting it
        const std::string msg = g_err;
        drain_streams(r, st);
        g_err = msg;
        return rc;
    }
    HIPCHK(hipEventRecord(r->ev_end, st));
    r->pending_stream = st;
    r->frame_pending = true;
    return RT_OK;
}

int frame_end(rt_renderer* r, rt_stats* stats) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    if (!r->frame_pending) return fail(RT_ERR_INVALID, "no frame in flight (rt_render_frame_begin)");
    HIPCHK(hipSetDevice(r->scene->device));
    hipStream_t st = r->pending_stream;
    rt_stats local = r->pending_stats;
    const size_t n_hot = r->pending_hot;
    r->frame_pending = false;
    HIPCHK(hipStreamSynchronize(st));
    const auto t1 = std::chrono::high_resolution_clock::now();
    local.seconds = std::chrono::duration<double>(t1 - r->pending_t0).count();
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, r->ev_begin, r->ev_end));
    local.device_ms = ms;
    if (r->kind == RT_RENDERER_MEGAKERNEL) {
        for (size_t i = 0; i + 1 < n_hot; i += 2) {
            float k = 0.0f;
            HIPCHK(hipEventElapsedTime(&k, r->ev_pool[i], r->ev_pool[i + 1]));
            local.hot_kernel_ms += k;
            local.hot_kernel_launches++;
            local.kernel_ms[r->max_depth == 0 ? RT_K_FILL_BLACK : RT_K_MEGAKERNEL] += k;
        }
    } else {
        // the dominant kernel of the schedule that ran: k_wf_extend where every bounce is a launch pair, k_wf_finish otherwise
        const int hot = std::min(r->max_depth, r->sched.finish_depth) >= r->max_depth && r->max_depth > 0 ? (r->sched.fused_bounce ? RT_K_WF_SHOOT : RT_K_WF_EXTEND) : RT_K_WF_FINISH;
        for (auto& L : r->lanes)
            for (const auto& sp : L.spans) {
                float k = 0.0f;
                HIPCHK(hipEventElapsedTime(&k, L.ev_pool[sp.e0], L.ev_pool[sp.e1]));
                local.kernel_ms[sp.kernel] += k;
                if (sp.kernel == hot) local.hot_kernel_ms += k, local.hot_kernel_launches++;
            }
    }
    if (r->n_local) {
        if (r->kind == RT_RENDERER_MEGAKERNEL) {
            unsigned long long ctr[8] = {0};
            HIPCHK(hipMemcpy(ctr, r->d_ray_counter, sizeof(ctr), hipMemcpyDeviceToHost));
            local.rays = ctr[0];
        } else {
            uint64_t rays = 0;
            const uint32_t D = r->max_depth;
            for (auto& L : r->lanes) {
                if (!L.n_local) continue;
                HIPCHK(hipMemcpy(L.h_counts.data(), L.d_counts, L.h_counts.size() * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(L.h_extra.data(), L.d_extra, L.h_extra.size() * 8, hipMemcpyDeviceToHost));
                if (L.h_extra[r->spp]) return fail(RT_ERR_HIP, "k_wf_finish: a queue hand-over stalled or overflowed (internal error word " + std::to_string((unsigned long long)L.h_extra[r->spp]) + ": 1 / 2 dynamic queue, 3 SHOOT's slot allocator)");
                const bool shoot_frame = r->sched.fused_bounce && std::min(D, r->sched.finish_depth) > 0; // SHOOT queues have holes: counted on the device
                for (uint32_t s = 0; s < r->spp; ++s) {
                    if (!shoot_frame)
                        for (uint32_t d = 0; d < D; ++d) rays += L.h_counts[(size_t)s * (D + 1) + d]; // :407 (queue lengths)
                    rays += L.h_extra[s];                                                            // rays k_wf_finish traced beyond a queue
                }
            }
            local.rays = rays;
#ifdef RT_DEBUG_COUNTS
            if (rays > (uint64_t)r->n_local * r->spp * r->max_depth) {
                for (auto& L : r->lanes) {
                    std::fprintf(stderr, "[debug] counts:");
                    for (size_t i = 0; i < L.h_counts.size(); ++i) std::fprintf(stderr, " %u", L.h_counts[i]);
                    std::fprintf(stderr, "\n[debug] extra:");
                    for (size_t i = 0; i < L.h_extra.size(); ++i) std::fprintf(stderr, " %llu", (unsigned long long)L.h_extra[i]);
                    std::fprintf(stderr, "\n");
                }
            }
#endif
        }
    }
    if (r->d_stats && r->n_local) {
        unsigned long long h[24];
        const unsigned long long rays = local.rays;
        HIPCHK(hipMemcpy(h, r->d_stats, sizeof(h), hipMemcpyDeviceToHost));
        std::fprintf(stderr,
                     "[rt stats] rays %llu | inner steps %llu (%.1f lanes/step) | leaf steps %llu (%.1f lanes/step) | "
                     "shade/refill rounds %llu (%.1f lanes/round) | live lanes/step %.1f | per ray: %.2f inner (%.2f in the top 341 nodes), %.2f tri\n",
                     rays, h[0], h[0] ? (double)h[1] / h[0] : 0.0, h[2], h[2] ? (double)h[3] / h[2] : 0.0, h[4],
                     h[4] ? (double)h[5] / h[4] : 0.0, (h[0] + h[2]) ? (double)h[6] / (h[0] + h[2]) : 0.0,
                     rays ? (double)h[1] / rays : 0.0, rays ? (double)h[7] / rays : 0.0, rays ? (double)h[3] / rays : 0.0);
        if (h[10])
            std::fprintf(stderr, "[rt stats] waves %llu | cycles per wave: mean %.0f, max %llu | traversal %.1f%%, shading %.1f%% of wave time | "
                         "%.0f cycles per traversal step, %.0f per shade round\n",
                         h[10], (double)h[8] / h[10], h[9], 100.0 * h[11] / h[8], 100.0 * h[12] / h[8],
                         (h[0] + h[2]) ? (double)h[11] / (h[0] + h[2]) : 0.0, h[4] ? (double)h[12] / h[4] : 0.0);
        if (h[10] && h[4] && (h[16] | h[17] | h[18] | h[19] | h[20])) // RT_ROUND_STAMPS builds
            std::fprintf(stderr, "[rt stats] shade round, cycles between stamps: record arrived %.0f | material + texel issued %.0f | normal %.0f | scatter + texel arrived %.0f | rest of the round %.0f\n",
                         (double)h[16] / h[4], (double)h[17] / h[4], (double)h[18] / h[4], (double)h[19] / h[4], (double)h[20] / h[4]);
        if (h[10] && h[15] > h[13] && h[14] != ~0ull)
            std::fprintf(stderr, "[rt stats] frame timeline: %.2f ms in all, pixel queue empty after %.2f ms, drain %.2f ms (%.1f%%)\n",
                         (h[15] - h[13]) * 1e-5, (h[14] - h[13]) * 1e-5, (h[15] - h[14]) * 1e-5, 100.0 * (h[15] - h[14]) / (h[15] - h[13]));
    }
    if (stats) *stats = local;
    return RT_OK;
}

int render_impl(rt_renderer* r, const rt_camera* cam, float* d_f32, uint8_t* d_u8, hipStream_t st, rt_stats* stats) {
    const int rc = frame_begin(r, cam, d_f32, d_u8, st);
    return rc != RT_OK ? rc : frame_end(r, stats);
}

} // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

// Camera::Camera (src/camera.hpp:74-106). Host arithmetic, same operation order as the reference's
// constructor: normalize, two cross products, viewport (aspect, 1), pixel00, per-pixel deltas.
int rt_camera_init(rt_camera* out, int32_t width, int32_t height, const float center[3], const float dir_in[3],
                   float focal_length) {
    if (!out || !center || !dir_in || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad camera arguments");
    auto norm = [](const float v[3], float o[3]) {
        float inv = 1.0f / std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        o[0] = v[0] * inv, o[1] = v[1] * inv, o[2] = v[2] * inv;
    };
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
    };
    float dir[3], right[3], up[3], tmp[3];
    const float world_up[3] = {0.0f, 1.0f, 0.0f};
    norm(dir_in, dir);
    cross(dir, world_up, tmp), norm(tmp, right);
    cross(right, dir, tmp), norm(tmp, up);
    const float vp0 = 1.0f * ((float)width / (float)height), vp1 = 1.0f;
    const float du_div = (float)width / (vp0 * 2.0f), dv_div = (float)height / (vp1 * 2.0f);
    for (int a = 0; a < 3; ++a) {
        const float viewport_u = (-right[a]) * vp0, viewport_v = up[a] * vp1;
        out->center[a] = center[a];
        out->pixel00[a] = ((center[a] + viewport_u) + viewport_v) + dir[a] * focal_length;
        out->delta_u[a] = right[a] / du_div;
        out->delta_v[a] = (-up[a]) / dv_div;
    }
    out->width = width, out->height = height;
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc* desc, int device, int bvh_kind, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (bvh_kind != RT_BVH_DEFAULT && bvh_kind != RT_BVH_LBVH && bvh_kind != RT_BVH_SAH && bvh_kind != RT_BVH_LBVH_GPU)
        return fail(RT_ERR_INVALID, "unknown bvh_kind");
    if (bvh_kind == RT_BVH_LBVH_GPU) { // the build itself runs on the device
        if (device < 0) return fail(RT_ERR_NO_DEVICE, "RT_BVH_LBVH_GPU needs a device (device >= 0)");
        int rc0 = device_ok(device);
        if (rc0 != RT_OK) return rc0;
    }
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_OOM, "host allocation failed");
    std::string err;
    int rc = RT_OK;
    try {
        rc = build_host_scene(desc, bvh_kind, s->hs, err);
    } catch (const std::bad_alloc&) {
        delete s;
        return fail(RT_ERR_OOM, "host allocation failed while building the scene");
    } catch (const std::exception& e) { // nothing may cross the C ABI as an exception
        delete s;
        return fail(RT_ERR_INVALID, std::string("scene build failed: ") + e.what());
    }
    if (rc != RT_OK) {
        delete s;
        return fail(rc, err);
    }
    s->device = device;
    if (device >= 0) {
        rc = device_ok(device);
        if (rc == RT_OK) {
            if (s->hs.nodes.size() > (size_t)(0x7FFFFFFF / 64)) { // (a tree of 33 M nodes: ~130 M triangles)
                rt_scene_destroy(s);
                return fail(RT_ERR_INVALID, "BVH too large for 32-bit node offsets");
            }
            std::vector<BvhNode> dn(s->hs.nodes);
            for (BvhNode& n : dn)
                for (int k = 0; k < 4; ++k)
                    if (n.child[k] >= 0) n.child[k] *= 64;
            rc = upload(dn, &s->dev.nodes, s->device_bytes);
        }
        if (rc == RT_OK) { // the device's triangle records: the host records' ten live dwords at a kTriBytes stride
            std::vector<uint8_t> packed(s->hs.tris.size() * (size_t)kTriBytes + 48, 0); // + 48: the whole-leaf step reads 80 bytes at every leaf, also at a last leaf of one record
            for (size_t i = 0; i < s->hs.tris.size(); ++i) std::memcpy(packed.data() + i * kTriBytes, &s->hs.tris[i], std::min<size_t>(kTriBytes, sizeof(TriRec)));
            rc = upload(packed, &s->dev.tris, s->device_bytes);
        }
        if (rc == RT_OK) rc = upload(s->hs.shade, &s->dev.shade, s->device_bytes);
        if (rc == RT_OK) rc = upload(s->hs.inst, &s->dev.inst, s->device_bytes);
        if (rc == RT_OK) rc = upload(s->hs.mats, &s->dev.mats, s->device_bytes);
        if (rc == RT_OK) rc = upload(s->hs.tex, &s->dev.tex, s->device_bytes);
        if (rc != RT_OK) {
            rt_scene_destroy(s);
            return rc;
        }
        s->dev.n_nodes = (uint32_t)s->hs.nodes.size();
        s->dev.n_tris = (uint32_t)(s->hs.wverts.size() / 9);
        std::memcpy(s->dev.sky, s->hs.sky, 12);
        s->dev.packed_mat = s->hs.packed_mat ? 1u : 0u;
        // what the shading kernels stage in LDS (rt_device.h: ShadeTables): the head of the distinct-matrix table and of the material table
        const bool stage = s->hs.packed_mat && s->hs.n_layers <= 65536u; // (the staged material keeps its layer in 16 bits)
        s->dev.lds_nm = stage ? (uint32_t)std::min<size_t>(s->hs.inst.size(), kLdsNm) : 0u;
        s->dev.lds_mats = stage ? (uint32_t)std::min<size_t>(s->hs.mats.size(), kLdsMats) : 0u;
        for (int a = 0; a < 3; ++a) {
            const float ext = s->hs.bounds_hi[a] - s->hs.bounds_lo[a];
            s->dev.cell_lo[a] = s->hs.bounds_lo[a];
            s->dev.cell_scale[a] = ext > 0.0f && std::isfinite(ext) ? 4.0f / ext : 0.0f;
        }
    }
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    if (s->device >= 0 && hipSetDevice(s->device) == hipSuccess) {
        (void)hipFree((void*)s->dev.nodes), (void)hipFree((void*)s->dev.tris), (void)hipFree((void*)s->dev.shade);
        (void)hipFree((void*)s->dev.inst), (void)hipFree((void*)s->dev.mats), (void)hipFree((void*)s->dev.tex);
    }
    delete s;
}

int rt_scene_info(const rt_scene* s, rt_scene_info_t* out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    out->n_triangles = (uint32_t)(s->hs.wverts.size() / 9);
    out->n_nodes = (uint32_t)s->hs.nodes.size();
    out->max_depth = s->hs.max_depth;
    out->max_leaf_tris = s->hs.max_leaf_tris;
    std::memcpy(out->bounds_lo, s->hs.bounds_lo, 12), std::memcpy(out->bounds_hi, s->hs.bounds_hi, 12);
    out->sah_cost = s->hs.sah_cost;
    out->device_bytes = s->device_bytes;
    out->n_leaf_records = (uint32_t)s->hs.tris.size();
    out->n_split_triangles = s->hs.n_split_triangles;
    return RT_OK;
}

int rt_scene_check_bvh(const rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    std::string err;
    int rc = check_bvh(s->hs, err);
    return rc == RT_OK ? RT_OK : fail(rc, err);
}

int rt_scene_count_visits(const rt_scene* s, uint32_t n, const float* org, const float* dir, int mode, uint64_t* node_visits, uint64_t* tri_tests,
                          float* t, uint32_t* tri) {
    if (!s || (n && (!org || !dir))) return fail(RT_ERR_INVALID, "null argument");
    std::string err;
    const int rc = no_throw([&] { return count_visits(s->hs, n, org, dir, mode, node_visits, tri_tests, t, tri, err); });
    return rc == RT_OK ? RT_OK : fail(rc, err.empty() ? g_err : err);
}

int rt_intersect_batch(rt_scene* s, uint32_t n, const float* org, const float* dir, float* t, float* u, float* v,
                       uint32_t* tri) {
    if (!s || (n && (!org || !dir || !t || !u || !v || !tri))) return fail(RT_ERR_INVALID, "null argument");
    if (s->device < 0) return fail(RT_ERR_NO_DEVICE, "scene was built host-only (device < 0)");
    if (n == 0) return RT_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (!origin_in_contract_range(s->hs, org + 3 * (size_t)i))
            return fail(RT_ERR_INVALID, "ray " + std::to_string(i) + ": origin more than 100 scene scales outside the scene's bounds (outside the range of the closest-hit contract)");
    HIPCHK(hipSetDevice(s->device));
    DevBuf b_org, b_dir, b_out;
    HIPCHK(b_org.alloc((size_t)n * 12));
    HIPCHK(b_dir.alloc((size_t)n * 12));
    HIPCHK(b_out.alloc((size_t)n * 16));
    float *d_org = b_org.as<float>(), *d_dir = b_dir.as<float>(), *d_out = b_out.as<float>();
    HIPCHK(hipMemcpy(d_org, org, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_dir, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    float* d_t = d_out;
    float* d_u = d_out + n;
    float* d_v = d_out + 2 * (size_t)n;
    uint32_t* d_tri = (uint32_t*)(d_out + 3 * (size_t)n);
    hipLaunchKernelGGL(k_intersect_batch, dim3((n + 255u) / 256u), dim3(256), 0, 0, s->dev, n, d_org, d_dir, d_t, d_u, d_v, d_tri);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(u, d_u, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(v, d_v, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_renderer_create(int kind, rt_scene* scene, int32_t width, int32_t height, uint32_t max_depth,
                       uint32_t sample_count, uint32_t seed_mode, rt_renderer** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (!scene) return fail(RT_ERR_INVALID, "null scene");
    if (kind != RT_RENDERER_MEGAKERNEL && kind != RT_RENDERER_WAVEFRONT) return fail(RT_ERR_INVALID, "unknown renderer kind");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(RT_ERR_INVALID, "bad image size");
    if (sample_count == 0) return fail(RT_ERR_INVALID, "sample_count must be >= 1");
    if (seed_mode > RT_SEED_MEGAKERNEL) return fail(RT_ERR_INVALID, "unknown seed mode");
    if ((uint64_t)sample_count * ((uint64_t)max_depth + 1) > (1ull << 26)) return fail(RT_ERR_INVALID, "sample_count * max_depth too large");
    if (scene->device < 0) return fail(RT_ERR_NO_DEVICE, "scene was built host-only (device < 0)");
    int rc = device_ok(scene->device);
    if (rc != RT_OK) return rc;
    rt_renderer* r = new (std::nothrow) rt_renderer();
    if (!r) return fail(RT_ERR_OOM, "host allocation failed");
    r->kind = kind, r->scene = scene, r->width = width, r->height = height;
    r->max_depth = max_depth, r->spp = sample_count;
    r->seed_mode = seed_mode != RT_SEED_DEFAULT ? seed_mode
                   : (kind == RT_RENDERER_MEGAKERNEL ? RT_SEED_MEGAKERNEL : RT_SEED_WAVEFRONT);
    r->hw_queues = hw_queues_from_env();
    const char* prof = std::getenv("RT_PROFILE_KERNELS");
    r->profile_kernels = prof && prof[0] == '1';
    if (const char* e = dev_knob("RT_MEGA_LDS_PAD")) r->mega_lds_pad = (uint32_t)std::max(0, std::min(100 * 1024, std::atoi(e)));
    if (const char* e = dev_knob("RT_MEGA_OCC")) r->mega_occ = (uint32_t)std::max(1, std::min((int)kMegaWaves, std::atoi(e)));
    auto bail = [&](int code) {
        rt_renderer_destroy(r);
        return code;
    };
    { // the device's lane-stream pool is set up by the FIRST renderer of either kind (see lane_stream_of: early, and in one burst)
        hipStream_t s0 = nullptr;
        bool owned = false;
        if (lane_stream_of(scene->device, 0, &s0, &owned) != hipSuccess) return bail(fail(RT_ERR_HIP, "cannot create the stream-lane pool"));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, scene->device) == hipSuccess && prop.multiProcessorCount > 0) r->n_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipStreamCreate failed"));
    if (hipEventCreate(&r->ev_begin) != hipSuccess || hipEventCreate(&r->ev_end) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->ev_tile_read, hipEventDisableTiming) != hipSuccess)
        return bail(fail(RT_ERR_HIP, "hipEventCreate failed"));
    if (kind == RT_RENDERER_MEGAKERNEL) {
        if (hipMalloc((void**)&r->d_frame, sizeof(MegaFrame)) != hipSuccess || hipHostMalloc((void**)&r->h_frame, sizeof(MegaFrame)) != hipSuccess)
            return bail(fail(RT_ERR_OOM, "frame constant buffer allocation failed"));
    }
    if (kind == RT_RENDERER_WAVEFRONT) {
        if (hipMalloc((void**)&r->d_cam, sizeof(CameraDev)) != hipSuccess || hipHostMalloc((void**)&r->h_cam, sizeof(CameraDev)) != hipSuccess)
            return bail(fail(RT_ERR_OOM, "camera buffer allocation failed"));
        // Environment shim for sweep scripts, DEVELOPER builds only (rt_knobs.h; the API is rt_renderer_set_schedule): the variables fill the initial schedule.
        // RT_WF_STREAMS / RT_WF_REQUEUE alone have always meant "not the one-launch schedule": they imply a launch per sample.
        rt_schedule& sc = r->sched;
        if (const char* e = dev_knob("RT_WF_GRAPH")) sc.hip_graph = e[0] == '1';
        if (const char* e = dev_knob("RT_WF_FINISH_DEPTH")) sc.finish_depth = (uint32_t)std::max(0, std::atoi(e));
        if (const char* e = dev_knob("RT_WF_SAMPLES_PER_LAUNCH")) sc.samples_per_launch = (uint32_t)std::max(0, std::atoi(e));
        if (const char* e = dev_knob("RT_WF_REQUEUE")) sc.requeue = e[0] != '0' ? 1 : 0;
        if (const char* e = dev_knob("RT_WF_STREAMS")) sc.stream_lanes = (uint32_t)std::max(1, std::min(8, std::atoi(e)));
        if ((dev_knob("RT_WF_REQUEUE") || dev_knob("RT_WF_STREAMS")) && sc.samples_per_launch == 0) sc.samples_per_launch = 1;
        if (const char* e = dev_knob("RT_WF_LPT")) sc.cost_order = e[0] == '0' ? 0 : (e[0] == '2' ? 1 : -1); // 2: forced
        if (const char* e = dev_knob("RT_WF_EXTEND_OCC")) r->wf_extend_occ = (uint32_t)std::max(1, std::min((int)kExtendWaves, std::atoi(e)));
        if (const char* e = dev_knob("RT_WF_FINISH_OCC")) r->wf_finish_occ = (uint32_t)std::max(1, std::min((int)kMegaWaves, std::atoi(e)));
        if (const char* e = dev_knob("RT_WF_SHOOT_CHUNK")) r->wf_shoot_chunk = (uint32_t)std::max(16, std::min(1024, std::atoi(e)));
        if (const char* e = dev_knob("RT_WF_SHOOT_TAIL")) r->wf_shoot_tail = e[0] == '1';
        if (const char* e = dev_knob("RT_WF_SHOOT_STATIC_PCT")) r->wf_shoot_static_pct = (uint32_t)std::max(0, std::min(100, std::atoi(e)));
        if (const char* e = dev_knob("RT_WF_REORDER")) sc.reorder = e[0] == '1';
        if (const char* e = dev_knob("RT_WF_MATSORT")) sc.matsort = e[0] == '1';
        if (const char* e = dev_knob("RT_WF_FUSED_BOUNCE")) sc.fused_bounce = e[0] == '1';
    }
    rc = no_throw([&] { return alloc_tile_buffers(r); });
    if (rc != RT_OK) return bail(rc);
    *out = r;
    return RT_OK;
}

void rt_renderer_destroy(rt_renderer* r) {
    if (!r) return;
    if (r->scene && hipSetDevice(r->scene->device) == hipSuccess) {
        drain_streams(r, r->pending_stream);
        free_tile_buffers(r);
        if (r->d_cam) (void)hipFree(r->d_cam);
        if (r->h_cam) (void)hipHostFree(r->h_cam);
        if (r->d_frame) (void)hipFree(r->d_frame);
        if (r->h_frame) (void)hipHostFree(r->h_frame);
        for (hipEvent_t e : r->ev_pool) (void)hipEventDestroy(e);
        if (r->ev_begin) (void)hipEventDestroy(r->ev_begin);
        if (r->ev_end) (void)hipEventDestroy(r->ev_end);
        if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
        if (r->ev_tile_read) (void)hipEventDestroy(r->ev_tile_read);
        if (r->stream) (void)hipStreamDestroy(r->stream);
    }
    delete r;
}

int rt_renderer_set_tile(rt_renderer* r, uint32_t rank, uint32_t world, uint32_t strip_rows) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    if (world == 0 || rank >= world || strip_rows == 0) return fail(RT_ERR_INVALID, "bad tile split");
    if (r->frame_pending) return fail(RT_ERR_INVALID, "a frame is in flight (rt_render_frame_end first)");
    HIPCHK(hipSetDevice(r->scene->device));
    drain_streams(r, r->pending_stream); // the queues are about to be freed: nothing may be running on any of the renderer's streams
    const TileDev old = r->tile;
    r->tile.rank = rank, r->tile.world = world, r->tile.strip_rows = strip_rows;
    const int rc = no_throw([&] { return alloc_tile_buffers(r); });
    if (rc != RT_OK) r->tile.rank = old.rank, r->tile.world = old.world, r->tile.strip_rows = old.strip_rows;
    return rc;
}

int32_t rt_renderer_local_rows(const rt_renderer* r) { return r ? r->tile.local_rows : 0; }

int32_t rt_renderer_global_row(const rt_renderer* r, int32_t local_row) {
    if (!r || local_row < 0 || local_row >= r->tile.local_rows) return -1;
    const uint32_t strip = (uint32_t)local_row / r->tile.strip_rows, within = (uint32_t)local_row % r->tile.strip_rows;
    return (int32_t)((strip * r->tile.world + r->tile.rank) * r->tile.strip_rows + within);
}

int rt_renderer_set_profiling(rt_renderer* r, int enable) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    r->profile_kernels = enable != 0;
    return RT_OK;
}

int rt_renderer_set_russian_roulette(rt_renderer* r, uint32_t start_bounce) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    if (r->rr_start != start_bounce) drop_graph(r); // the bounce flags are baked into the captured launches
    r->rr_start = start_bounce;
    return RT_OK;
}

int rt_renderer_get_schedule(const rt_renderer* r, rt_schedule* out) {
    if (!r || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = r->sched;
    return RT_OK;
}

int rt_renderer_set_schedule(rt_renderer* r, const rt_schedule* s) {
    if (!r || !s) return fail(RT_ERR_INVALID, "null argument");
    if (s->stream_lanes > 8) return fail(RT_ERR_INVALID, "at most 8 stream lanes");
    if (s->requeue < -1 || s->requeue > 1 || s->cost_order < -1 || s->cost_order > 1 || s->reorder > 1 || s->matsort > 1 || s->hip_graph > 1 || s->fused_bounce > 1)
        return fail(RT_ERR_INVALID, "schedule field out of range");
    if (r->frame_pending) return fail(RT_ERR_INVALID, "a frame is in flight (rt_render_frame_end first)");
    if (s->pixel_slices < -1 || s->pixel_slices > (int32_t)kMaxSlices) return fail(RT_ERR_INVALID, "pixel_slices: -1 (automatic), 0 or 1 (off), 2 .. 8");
    if (r->kind != RT_RENDERER_WAVEFRONT && s->pixel_slices == r->sched.pixel_slices) { // the megakernel is one launch: only its pixel slices are a choice
        r->sched = *s;
        return RT_OK;
    }
    HIPCHK(hipSetDevice(r->scene->device));
    drain_streams(r, r->pending_stream); // the queues are re-allocated (stream lanes, second queue, hit records, dynamic queue)
    const rt_schedule old = r->sched;
    r->sched = *s;
    const int rc = no_throw([&] { return alloc_tile_buffers(r); });
    if (rc != RT_OK) r->sched = old; // the buffers are gone (frames are refused), the schedule on record is the last one that worked
    return rc;
}

int rt_render_frame(rt_renderer* r, const rt_camera* cam, float* rgba_f32, uint8_t* rgba_u8, rt_stats* stats) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    int rc = no_throw([&] { return render_impl(r, cam, rgba_f32 ? r->d_f32 : nullptr, rgba_u8 ? r->d_u8 : nullptr, r->stream, stats); });
    if (rc != RT_OK) return rc;
    if (rgba_f32 && r->n_local) HIPCHK(hipMemcpy(rgba_f32, r->d_f32, (size_t)r->n_local * 16, hipMemcpyDeviceToHost));
    if (rgba_u8 && r->n_local) HIPCHK(hipMemcpy(rgba_u8, r->d_u8, (size_t)r->n_local * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render_frame_device(rt_renderer* r, const rt_camera* cam, void* d_rgba_f32, void* d_rgba_u8, void* stream,
                           rt_stats* stats) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    return no_throw([&] { return render_impl(r, cam, (float*)d_rgba_f32, (uint8_t*)d_rgba_u8, stream ? (hipStream_t)stream : r->stream, stats); });
}

int rt_render_frame_begin(rt_renderer* r, const rt_camera* cam, void* d_rgba_f32, void* d_rgba_u8, void* stream) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    return no_throw([&] { return frame_begin(r, cam, (float*)d_rgba_f32, (uint8_t*)d_rgba_u8, stream ? (hipStream_t)stream : r->stream); });
}

int rt_render_frame_end(rt_renderer* r, rt_stats* stats) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    return no_throw([&] { return frame_end(r, stats); });
}

} // extern "C"

// ---- multi-GPU frame gather: ncclGather of the strip buffers + de-interleave on the root -------------------
namespace {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err; // why the library is unusable: dlerror() read ONCE, right after the failing call (a second dlerror() returns NULL)
    bool ok() const { return lib && CommInitAll && CommDestroy && GroupStart && GroupEnd && Gather && GetErrorString; }
};

RcclApi& rccl() { // loaded once, on the first rt_comm_create that needs it
    static RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
            const char* e = dlerror();
            a.err = e ? e : "dlopen failed";
        }
        if (a.lib) {
            a.err.clear();
            auto sym = [&a](const char* name) {
                void* f = dlsym(a.lib, name);
                if (!f && a.err.empty()) {
                    const char* e = dlerror();
                    a.err = e ? e : (std::string("symbol missing: ") + name);
                }
                return f;
            };
            a.CommInitAll = (decltype(a.CommInitAll))sym("ncclCommInitAll");
            a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
            a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
            a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
            a.Gather = (decltype(a.Gather))sym("ncclGather");
            a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        }
        return a;
    }();
    return api;
}

#define NCCLCHK(expr)                                                                                     \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

// gathered[rank][row][x] (every rank padded to rows_max rows) -> full[global row][x]; one thread per pixel of the frame
__global__ void __launch_bounds__(256) k_deinterleave(uint32_t width, uint32_t height, uint32_t world, uint32_t strip_rows, uint32_t rows_max,
                                                       const float4* __restrict__ g_f32, const uint32_t* __restrict__ g_u8,
                                                       float4* __restrict__ full_f32, uint32_t* __restrict__ full_u8) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t x = i % width, y = i / width;
    const uint32_t strip = y / strip_rows, rank = strip % world;
    const uint32_t local_row = (strip / world) * strip_rows + y % strip_rows; // the inverse of tile_global_row
    const size_t src = ((size_t)rank * rows_max + local_row) * width + x;
    if (g_f32) full_f32[i] = g_f32[src];
    if (g_u8) full_u8[i] = g_u8[src];
}

} // namespace

struct rt_comm {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms; // empty: local mode (a device appears more than once)
    bool use_rccl = false;
    // root-device buffers, (re)allocated when the frame geometry changes
    int32_t width = 0, height = 0;
    uint32_t rows_max = 0;
    float4* d_gather_f32 = nullptr;
    uint32_t* d_gather_u8 = nullptr;
    float4* d_full_f32 = nullptr;
    uint32_t* d_full_u8 = nullptr;
    hipStream_t root_stream = nullptr;
    // ordering without host waits: ev_tile[i] (on device i) marks "tile i's strips have left / arrived" on renderer i's stream,
    // ev_frame (root) marks "the de-interleave has read the gather buffer": the next gather's receive waits for it
    std::vector<hipEvent_t> ev_tile;
    hipEvent_t ev_frame = nullptr;
    bool frame_recorded = false;
    bool gathered_f32 = false, gathered_u8 = false; // planes of the last rt_frame_gather_begin
    bool broken = false; // a collective failed half-way: the communicators are unusable (rt_comm_destroy + rt_comm_create)
};

namespace {
void comm_free_buffers(rt_comm* c) {
    auto fr = [](auto*& p) {
        if (p) (void)hipFree((void*)p);
        p = nullptr;
    };
    fr(c->d_gather_f32), fr(c->d_gather_u8), fr(c->d_full_f32), fr(c->d_full_u8);
}
} // namespace

extern "C" {

int rt_comm_create(int n, const int* devices, rt_comm** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (n <= 0 || n > 64 || !devices) return fail(RT_ERR_INVALID, "bad device list");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device available");
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= n_dev) return fail(RT_ERR_INVALID, "device index out of range");
        for (int j = 0; j < i; ++j) distinct = distinct && devices[j] != devices[i];
    }
    return no_throw([&] {
        rt_comm* c = new rt_comm();
        c->devices.assign(devices, devices + n);
        auto bail = [&](int code) {
            rt_comm_destroy(c);
            return code;
        };
        if (hipSetDevice(devices[0]) != hipSuccess || hipStreamCreateWithFlags(&c->root_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_frame, hipEventDisableTiming) != hipSuccess)
            return bail(fail(RT_ERR_HIP, "cannot create a stream on the root device"));
        c->ev_tile.assign((size_t)n, nullptr);
        for (int i = 0; i < n; ++i)
            if (hipSetDevice(devices[i]) != hipSuccess || hipEventCreateWithFlags(&c->ev_tile[(size_t)i], hipEventDisableTiming) != hipSuccess)
                return bail(fail(RT_ERR_HIP, "cannot create an event on device " + std::to_string(devices[i])));
        (void)hipSetDevice(devices[0]);
        if (distinct) { // one rank per device: RCCL over xGMI
            if (!rccl().ok()) return bail(fail(RT_ERR_UNSUPPORTED, "librccl.so could not be loaded: " + (rccl().err.empty() ? std::string("symbols missing") : rccl().err)));
            c->comms.assign((size_t)n, nullptr);
            const ncclResult_t rc = rccl().CommInitAll(c->comms.data(), n, devices);
            if (rc != ncclSuccess) {
                c->comms.clear();
                return bail(fail(RT_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(rc)));
            }
            c->use_rccl = true;
        }
        *out = c;
        return (int)RT_OK;
    });
}

void rt_comm_destroy(rt_comm* c) {
    if (!c) return;
    for (size_t i = 0; i < c->comms.size(); ++i)
        if (c->comms[i] && hipSetDevice(c->devices[i]) == hipSuccess) (void)rccl().CommDestroy(c->comms[i]);
    for (size_t i = 0; i < c->ev_tile.size(); ++i)
        if (c->ev_tile[i] && hipSetDevice(c->devices[i]) == hipSuccess) (void)hipEventDestroy(c->ev_tile[i]);
    if (!c->devices.empty() && hipSetDevice(c->devices[0]) == hipSuccess) {
        if (c->root_stream) (void)hipStreamSynchronize(c->root_stream);
        comm_free_buffers(c);
        if (c->ev_frame) (void)hipEventDestroy(c->ev_frame);
        if (c->root_stream) (void)hipStreamDestroy(c->root_stream);
    }
    delete c;
}

int rt_comm_uses_rccl(const rt_comm* c) { return c && c->use_rccl ? 1 : 0; }
const void* rt_comm_frame_f32(const rt_comm* c) { return c ? c->d_full_f32 : nullptr; }
const void* rt_comm_frame_u8(const rt_comm* c) { return c ? c->d_full_u8 : nullptr; }
void* rt_renderer_tile_f32(rt_renderer* r) { return r ? r->d_f32 : nullptr; }
void* rt_renderer_tile_u8(rt_renderer* r) { return r ? r->d_u8 : nullptr; }

// Enqueues the gather of the renderers' last frames and returns at once: nothing in here waits on the host. Order on the device:
//   renderer i's stream : [frame i rendered] -> (root only: wait ev_frame = the previous de-interleave has read the gather buffer)
//                         -> ncclGather send (root: + receive) -> record ev_tile[i]
//   root stream         : wait ev_tile[0] (RCCL: the receive is complete when the root's part of the collective is; device copies:
//                         wait every ev_tile[i], then the copies) -> k_deinterleave -> record ev_frame
// A renderer's next frame is enqueued on its own stream behind its send, so the caller may begin the next frame right after this
// call and collect this one later with rt_comm_wait: the gather overlaps with the next frame's traversal.
int rt_frame_gather_begin(rt_comm* c, rt_renderer* const* rs, int want_f32, int want_u8) {
    if (!c || !rs) return fail(RT_ERR_INVALID, "null communicator or renderer list");
    if (c->broken) return fail(RT_ERR_HIP, "a collective of this communicator failed: destroy it and create a new one");
    const uint32_t n = (uint32_t)c->devices.size();
    const bool f32 = want_f32 != 0, u8 = want_u8 != 0;
    if (!f32 && !u8) return fail(RT_ERR_INVALID, "nothing to gather: ask for the f32 and / or the u8 plane");
    for (uint32_t i = 0; i < n; ++i) {
        const rt_renderer* r = rs[i];
        if (!r) return fail(RT_ERR_INVALID, "null renderer");
        if (r->frame_pending) return fail(RT_ERR_INVALID, "a renderer still has a frame in flight (rt_render_frame_end first)");
        if (r->unusable) return fail(RT_ERR_INVALID, "renderer " + std::to_string(i) + " has no buffers: its last rt_renderer_set_tile / rt_renderer_set_schedule failed (call either again)");
        if (r->scene->device != c->devices[i]) return fail(RT_ERR_INVALID, "renderer " + std::to_string(i) + " is not on the communicator's device " + std::to_string(c->devices[i]));
        if (r->width != rs[0]->width || r->height != rs[0]->height) return fail(RT_ERR_INVALID, "renderers differ in frame size");
        if (r->tile.world != n || r->tile.rank != i || r->tile.strip_rows != rs[0]->tile.strip_rows)
            return fail(RT_ERR_INVALID, "renderer " + std::to_string(i) + " is not tile (" + std::to_string(i) + ", " + std::to_string(n) + ") of the frame");
    }
    return no_throw([&]() -> int {
        const int32_t W = rs[0]->width, H = rs[0]->height;
        const uint32_t strip = rs[0]->tile.strip_rows, rows_max = (uint32_t)rs[0]->tile.local_rows; // rank 0 owns the most rows
        const size_t tile_px = (size_t)rows_max * (size_t)W, frame_px = (size_t)W * (size_t)H;
        HIPCHK(hipSetDevice(c->devices[0]));
        if (c->width != W || c->height != H || c->rows_max != rows_max) {
            HIPCHK(hipStreamSynchronize(c->root_stream));
            comm_free_buffers(c);
            c->width = W, c->height = H, c->rows_max = rows_max;
            c->frame_recorded = false;
        }
        if (f32 && !c->d_full_f32) {
            HIPCHK(hipMalloc((void**)&c->d_gather_f32, std::max<size_t>(1, tile_px * n) * 16));
            HIPCHK(hipMalloc((void**)&c->d_full_f32, std::max<size_t>(1, frame_px) * 16));
        }
        if (u8 && !c->d_full_u8) {
            HIPCHK(hipMalloc((void**)&c->d_gather_u8, std::max<size_t>(1, tile_px * n) * 4));
            HIPCHK(hipMalloc((void**)&c->d_full_u8, std::max<size_t>(1, frame_px) * 4));
        }
        c->gathered_f32 = f32, c->gathered_u8 = u8;
        if (frame_px == 0) return RT_OK;
        if (c->use_rccl) {
            // the previous frame's de-interleave must have read the gather buffer before the root receives into it again
            if (c->frame_recorded) HIPCHK(hipStreamWaitEvent(rs[0]->stream, c->ev_frame, 0));
            // ONE grouped collective per plane: every rank sends rows_max * W pixels from its own stream (ordered behind its
            // render), the root receives n blocks. Single-process multi-device RCCL calls must be grouped (rccl.h:213). An error
            // inside the group is remembered and the group is CLOSED all the same: returning between GroupStart and GroupEnd would
            // leave this thread's group open and every later collective queued instead of issued.
            ncclResult_t first = rccl().GroupStart();
            if (first != ncclSuccess) return fail(RT_ERR_HIP, std::string("ncclGroupStart: ") + rccl().GetErrorString(first));
            std::string what;
            for (uint32_t i = 0; i < n && first == ncclSuccess; ++i) {
                if (hipSetDevice(c->devices[i]) != hipSuccess) {
                    first = ncclUnhandledCudaError, what = "hipSetDevice";
                    break;
                }
                if (f32) first = rccl().Gather(rs[i]->d_f32, i == 0 ? (void*)c->d_gather_f32 : nullptr, tile_px * 4, ncclFloat, 0, c->comms[i], rs[i]->stream), what = "ncclGather (f32)";
                if (u8 && first == ncclSuccess) first = rccl().Gather(rs[i]->d_u8, i == 0 ? (void*)c->d_gather_u8 : nullptr, tile_px * 4, ncclUint8, 0, c->comms[i], rs[i]->stream), what = "ncclGather (u8)";
            }
            const ncclResult_t ge = rccl().GroupEnd();
            (void)hipSetDevice(c->devices[0]);
            if (first != ncclSuccess || ge != ncclSuccess) {
                c->broken = true; // part of the collective may be queued on some ranks' streams and never complete
                return fail(RT_ERR_HIP, (first != ncclSuccess ? what : std::string("ncclGroupEnd")) + ": " + rccl().GetErrorString(first != ncclSuccess ? first : ge) +
                                            " (the communicator is unusable now: rt_comm_destroy + rt_comm_create)");
            }
            HIPCHK(hipEventRecord(c->ev_tile[0], rs[0]->stream)); // the root's receive
            HIPCHK(hipStreamWaitEvent(c->root_stream, c->ev_tile[0], 0));
        } else {
            // the same device more than once (rehearsal): RCCL allows one rank per device, so the blocks move by device copies
            for (uint32_t i = 0; i < n; ++i) {
                HIPCHK(hipSetDevice(c->devices[i]));
                HIPCHK(hipEventRecord(c->ev_tile[i], rs[i]->stream));
            }
            HIPCHK(hipSetDevice(c->devices[0]));
            for (uint32_t i = 0; i < n; ++i) {
                HIPCHK(hipStreamWaitEvent(c->root_stream, c->ev_tile[i], 0));
                if (f32) HIPCHK(hipMemcpyAsync(c->d_gather_f32 + tile_px * i, rs[i]->d_f32, tile_px * 16, hipMemcpyDeviceToDevice, c->root_stream));
                if (u8) HIPCHK(hipMemcpyAsync(c->d_gather_u8 + tile_px * i, rs[i]->d_u8, tile_px * 4, hipMemcpyDeviceToDevice, c->root_stream));
            }
        }
        hipLaunchKernelGGL(k_deinterleave, dim3((uint32_t)((frame_px + 255) / 256)), dim3(256), 0, c->root_stream, (uint32_t)W, (uint32_t)H, n, strip,
                           rows_max, f32 ? c->d_gather_f32 : nullptr, u8 ? c->d_gather_u8 : nullptr, c->d_full_f32, c->d_full_u8);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev_frame, c->root_stream));
        c->frame_recorded = true;
        // A renderer's next frame must not overwrite its tile before it has been read: RCCL's send sits on the renderer's own stream already;
        // the device copies read from the root stream, so the own stream is made to wait for them. Either way an event recorded on the own
        // stream now marks "tile read", and rt_render_frame_begin makes a CALLER's stream wait for it too.
        for (uint32_t i = 0; i < n; ++i) {
            HIPCHK(hipSetDevice(c->devices[i]));
            if (!c->use_rccl) HIPCHK(hipStreamWaitEvent(rs[i]->stream, c->ev_frame, 0));
            HIPCHK(hipEventRecord(rs[i]->ev_tile_read, rs[i]->stream));
            rs[i]->tile_read_recorded = true;
        }
        HIPCHK(hipSetDevice(c->devices[0]));
        return RT_OK;
    });
}

// Waits for the gather enqueued by rt_frame_gather_begin; copies the frame to the host where a pointer is given.
int rt_comm_wait(rt_comm* c, float* rgba_f32, uint8_t* rgba_u8) {
    if (!c) return fail(RT_ERR_INVALID, "null communicator");
    if ((rgba_f32 && !c->gathered_f32) || (rgba_u8 && !c->gathered_u8)) return fail(RT_ERR_INVALID, "that plane was not gathered");
    const size_t frame_px = (size_t)c->width * (size_t)c->height;
    HIPCHK(hipSetDevice(c->devices[0]));
    if (rgba_f32 && frame_px) HIPCHK(hipMemcpyAsync(rgba_f32, c->d_full_f32, frame_px * 16, hipMemcpyDeviceToHost, c->root_stream));
    if (rgba_u8 && frame_px) HIPCHK(hipMemcpyAsync(rgba_u8, c->d_full_u8, frame_px * 4, hipMemcpyDeviceToHost, c->root_stream));
    HIPCHK(hipStreamSynchronize(c->root_stream));
    return RT_OK;
}

int rt_frame_gather(rt_comm* c, rt_renderer* const* rs, float* rgba_f32, uint8_t* rgba_u8, int want_device_f32, int want_device_u8) {
    const int rc = rt_frame_gather_begin(c, rs, rgba_f32 || want_device_f32, rgba_u8 || want_device_u8);
    return rc != RT_OK ? rc : rt_comm_wait(c, rgba_f32, rgba_u8);
}

int rt_comm_size(const rt_comm* c) { return c ? (int)c->devices.size() : 0; }

} // extern "C"

extern "C" {

int rt_probe_xorshift(int device, uint32_t seed, uint32_t n, float* out, uint32_t* state_out) {
    if (!out || !state_out || n == 0) return fail(RT_ERR_INVALID, "bad arguments");
    int rc = device_ok(device);
    if (rc != RT_OK) return rc;
    DevBuf b_out, b_st;
    HIPCHK(b_out.alloc((size_t)n * 4));
    HIPCHK(b_st.alloc(4));
    hipLaunchKernelGGL(k_probe_xorshift, dim3(1), dim3(64), 0, 0, seed, n, b_out.as<float>(), b_st.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, b_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(state_out, b_st.p, 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_probe_half_roundtrip(int device, uint32_t n, const float* in, float* out, uint16_t* bits_out) {
    if (!in || !out || !bits_out || n == 0) return fail(RT_ERR_INVALID, "bad arguments");
    int rc = device_ok(device);
    if (rc != RT_OK) return rc;
    DevBuf b_in, b_out, b_bits;
    HIPCHK(b_in.alloc((size_t)n * 4));
    HIPCHK(b_out.alloc((size_t)n * 4));
    HIPCHK(b_bits.alloc((size_t)n * 2));
    HIPCHK(hipMemcpy(b_in.p, in, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_half, dim3((n + 255u) / 256u), dim3(256), 0, 0, n, b_in.as<float>(), b_out.as<float>(), b_bits.as<uint16_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, b_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(bits_out, b_bits.p, (size_t)n * 2, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_probe_rounding(int device, uint64_t* mismatches) {
    if (!mismatches) return fail(RT_ERR_INVALID, "bad arguments");
    int rc = device_ok(device);
    if (rc != RT_OK) return rc;
    DevBuf b;
    HIPCHK(b.alloc(16));
    HIPCHK(hipMemset(b.p, 0, 16));
    hipLaunchKernelGGL(k_probe_rounding, dim3(4096), dim3(256), 0, 0, b.as<unsigned long long>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(mismatches, b.p, 16, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_probe_scatter(rt_scene* s, uint32_t material, uint32_t n, const float* dir, const float* normal, const float* uv,
                     const uint32_t* seed, uint8_t* ok, float* out_dir, float* out_att, uint32_t* seed_out) {
    if (!s || !dir || !normal || !uv || !seed || !ok || !out_dir || !out_att || !seed_out || n == 0)
        return fail(RT_ERR_INVALID, "bad arguments");
    if (s->device < 0) return fail(RT_ERR_NO_DEVICE, "scene was built host-only (device < 0)");
    if (material >= s->hs.mats.size()) return fail(RT_ERR_INVALID, "material index out of range");
    HIPCHK(hipSetDevice(s->device));
    DevBuf b_dir, b_nrm, b_uv, b_od, b_oa, b_seed, b_so, b_ok;
    HIPCHK(b_dir.alloc((size_t)n * 12));
    HIPCHK(b_nrm.alloc((size_t)n * 12));
    HIPCHK(b_uv.alloc((size_t)n * 8));
    HIPCHK(b_od.alloc((size_t)n * 12));
    HIPCHK(b_oa.alloc((size_t)n * 12));
    HIPCHK(b_seed.alloc((size_t)n * 4));
    HIPCHK(b_so.alloc((size_t)n * 4));
    HIPCHK(b_ok.alloc((size_t)n));
    HIPCHK(hipMemcpy(b_dir.p, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_nrm.p, normal, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_uv.p, uv, (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_seed.p, seed, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_scatter, dim3((n + 255u) / 256u), dim3(256), 0, 0, s->dev, material, n, b_dir.as<float>(),
                       b_nrm.as<float>(), b_uv.as<float>(), b_seed.as<uint32_t>(), b_ok.as<uint8_t>(), b_od.as<float>(),
                       b_oa.as<float>(), b_so.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(ok, b_ok.p, (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_dir, b_od.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_att, b_oa.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(seed_out, b_so.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

} // extern "C"
