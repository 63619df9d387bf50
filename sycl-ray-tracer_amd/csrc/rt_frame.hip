// rt_frame.hip — the frames of librt_mi355x.so: tile buffers, schedules, and every launch of a frame (gfx950).
//
// Host orchestration of the two renderers. Unlike the reference, which blocks on the host after
// every kernel (769 blocking submits per 64-spp frame: src/render_wavefront.cpp:396-431), a frame
// here is ONE asynchronous stream of launches: queue lengths live in device memory, one counter
// per (sample, depth), so no bounce ever waits for the host.
#include "rt_internal.h"
#include "rt_kernels.h"


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

namespace rtlib {

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
    // automatic: pixel slices (round 5) do more for the same tiles — half the 1080p frame 59 ms against 69 with the cost-ordered second launch —
    // so the cost order is left to tiles that are not sliced
    const bool sliced = r->sched.pixel_slices > 1 || (r->sched.pixel_slices < 0 && (double)r->n_local > 1.25 * r->n_cus * 4.0 * kMegaWaves * 64.0 && r->spp >= 2);
    if (sliced && !r->sched.hip_graph) return false;
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
    const char* ks = std::getenv("RT_KERNEL_STATS"); // 1: the fully instrumented instantiations; 2: the megakernel's timing-only one (rt_kernels.h)
    r->stats_level = ks && ks[0] == '2' && r->kind == RT_RENDERER_MEGAKERNEL ? 1 : 2;
    if (ks && (ks[0] == '1' || ks[0] == '2')) HIPCHK(dalloc((void**)&r->d_stats, 24 * sizeof(unsigned long long)));
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
            if (K == 1 && wf_one_launch(r) && !r->sched.hip_graph && r->sched.pixel_slices != 0 && r->sched.pixel_slices != 1 && (uint64_t)m * 32u < (1ull << 31)) {
                // pixel slices of the one-launch schedule (k_wf_finish<.., SLICED>): the state that changes lanes, 32 bytes per pixel (a replayed
                // hipGraph would replay the frame's tags too: no slices there)
                HIPCHK(dalloc((void**)&r->d_slice_state, m * 32));
                HIPCHK(hipMemsetAsync(r->d_slice_state, 0, m * 32, r->stream));
                HIPCHK(hipStreamSynchronize(r->stream));
                r->slice_tag = 0;
            }
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
SliceDev mega_slices(const rt_renderer* r, uint32_t grid, uint32_t n_slots, uint32_t tag0) { // grid: workgroups of kMegaBlock threads the launch keeps resident
    SliceDev sl{};
    sl.state = r->d_slice_state, sl.n_slices = 1, sl.tag0 = tag0, sl.wait_cap = kSliceWaitCap;
    if (const char* e = dev_knob("RT_SLICE_WAIT_CAP")) sl.wait_cap = (uint32_t)std::strtoul(e, nullptr, 10); // tests: give up after that many looks
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
    // test hook (developer build): the cursor hands out the later slices, but no lane ever ends one — every lane that takes a later slice waits for a
    // state that is never written, which is exactly the logic error the kernels' bounded wait has to turn into an error code
    if (dev_knob("RT_INJECT_SLICE_LOSS") && sl.n_slices > 1) sl.cuts = 0;
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
                // pixel slices: the one-launch schedule on one lane, outside a graph capture and the cost-ordered pair of launches
                SliceDev sl{};
                sl.n_slices = 1;
                if (K == 1 && r->d_slice_state && !capturing && !lpt && n_samples == r->spp && D0 == 0 && !shoot_frame) {
                    if (r->slice_tag > 0xFFFFFFFFu - 2u * (r->spp + 1u)) {
                        HIPCHK(hipMemsetAsync(r->d_slice_state, 0, (size_t)L.n_local * 32, ls));
                        r->slice_tag = 0;
                    }
                    r->slice_tag += r->spp + 1u;
                    sl = mega_slices(r, fin_grid, L.n_local, r->slice_tag);
                    r->h_frame->cam = *r->h_cam, r->h_frame->sl = sl;
                    HIPCHK(hipMemcpyAsync(r->d_frame, r->h_frame, sizeof(MegaFrame), hipMemcpyHostToDevice, ls));
                }
                r->last_slices = sl.n_slices;
                FinishRequeue fq{L.d_rq_entries, L.d_rq_ctl, reinterpret_cast<uint32_t*>(L.d_extra + r->spp), L.rq_capacity};
                if (req) { // the launch's dynamic queue: all entries invalid, counters zero
                    if (int rc_ = fill_bytes(L.d_rq_entries, 0xFF, (size_t)L.n_local * (n_samples - 1u) * 8, ls, capturing)) return rc_;
                    if (int rc_ = fill_bytes(L.d_rq_ctl, 0, 80 * 4, ls, capturing)) return rc_;
                }
#define RT_FINISH_LAUNCH(ST, RQ, SL)                                                                                                          \
    hipLaunchKernelGGL((k_wf_finish<ST, RQ, false, SL>), dim3(fin_grid), dim3(kMegaBlock), 0, ls, S, r->d_cam, L.tile, L.q[buf[k]], counts + D0, \
                       cursors + (size_t)s * D + D0, L.d_rng, L.d_accum, extra, D0, D, n_samples, r->rr_start,                                 \
                       ST ? r->d_stats : (unsigned long long*)nullptr, fq, chunk_slots, lpt && s == 0 ? L.d_cost : (uint8_t*)nullptr, QueueDev{}, (uint32_t*)nullptr, 0u,     \
                       shoot_frame ? 1u : 0u, r->d_frame, sl.n_slices)
                if (sl.n_slices > 1u && r->d_stats) RT_FINISH_LAUNCH(true, false, true);
                else if (sl.n_slices > 1u) RT_FINISH_LAUNCH(false, false, true);
                else if (r->d_stats && req) RT_FINISH_LAUNCH(true, true, false);
                else if (r->d_stats) RT_FINISH_LAUNCH(true, false, false);
                else if (req) RT_FINISH_LAUNCH(false, true, false);
                else RT_FINISH_LAUNCH(false, false, false);
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
        uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>((tiles_x * tiles_y + mega_wpb - 1u) / mega_wpb,
                                                                  (uint32_t)r->n_cus * 4u * r->mega_occ / mega_wpb));
        if (const char* e = dev_knob("RT_MEGA_CHAIN_GRID_PCT")) // experiment: more (emptier) waves than the tile needs, in per cent of what it needs
            grid = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)((unsigned long long)grid * std::strtoul(e, nullptr, 10) / 100u), (uint32_t)r->n_cus * 4u * r->mega_occ / mega_wpb));
        const uint32_t pad = r->mega_lds_pad;
        if (pad) { // static + dynamic LDS may exceed the 64 KiB a launch gets without asking
            (void)hipFuncSetAttribute((const void*)k_megakernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            (void)hipFuncSetAttribute((const void*)k_megakernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            (void)hipFuncSetAttribute((const void*)k_megakernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            (void)hipFuncSetAttribute((const void*)k_megakernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            (void)hipFuncSetAttribute((const void*)k_megakernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            (void)hipFuncSetAttribute((const void*)k_megakernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
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
            if (r->stats_level == 1) { // timing only
                if (sl.n_slices > 1u) RT_MEGA_LAUNCH(1, true);
                else RT_MEGA_LAUNCH(1, false);
            } else if (sl.n_slices > 1u) RT_MEGA_LAUNCH(2, true);
            else RT_MEGA_LAUNCH(2, false);
        } else {
            if (sl.n_slices > 1u) RT_MEGA_LAUNCH(0, true);
            else RT_MEGA_LAUNCH(0, false);
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
            stats->pixel_slices = r->last_slices;
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
            if (ctr[2]) return fail(RT_ERR_HIP, "k_megakernel: a pixel slice waited for its pixel's state for longer than any hand-over takes (internal error)");
        } else {
            uint64_t rays = 0;
            const uint32_t D = r->max_depth;
            for (auto& L : r->lanes) {
                if (!L.n_local) continue;
                HIPCHK(hipMemcpy(L.h_counts.data(), L.d_counts, L.h_counts.size() * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(L.h_extra.data(), L.d_extra, L.h_extra.size() * 8, hipMemcpyDeviceToHost));
                if (L.h_extra[r->spp]) return fail(RT_ERR_HIP, "k_wf_finish: a queue hand-over stalled or overflowed (internal error word " + std::to_string((unsigned long long)L.h_extra[r->spp]) + ": 1 / 2 dynamic queue, 3 SHOOT's slot allocator, 4 pixel-slice hand-over)");
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
        if (r->kind == RT_RENDERER_MEGAKERNEL && r->stats_level == 1)
            std::fprintf(stderr, "[rt stats] timing-only instantiation: rays %llu | traversal steps %llu | shade/refill rounds %llu\n", rays, h[0], h[4]);
        else
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

} // namespace rtlib
