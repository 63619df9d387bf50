// rt_kernel_stream.h — the per-bounce wavefront with OVERLAPPED launches (rt_schedule::fused_bounce = 2).
//
// The reference launches shoot_rays once per bounce (src/render_wavefront.cpp:396-417) and waits for it; k_wf_finish<.., LIMIT> ("SHOOT")
// keeps that shape — one kernel per bounce that intersects, shades and compacts the survivors into the next bounce's queue — but a
// launch cannot be shorter than its slowest ray, and the next one cannot start before it ends: 0.28 ms per launch for 0.21 ms of work
// (EXPERIMENTS.md). Here the launch of bounce b + 1 runs WHILE bounce b is still producing its queue:
//   * bounce b's launch sits on one HIP stream, bounce b + 1's on another, b + 2's behind b on the first: at most two are resident, each
//     asking for half of the SIMDs' wave slots, so both always fit (no launch ever waits for room another one holds);
//   * a queue slot carries its own validity: the last store of an entry is an 8-byte word {tag of the producing launch, the pixel's RNG
//     word}, written after the entry's other planes have been retired (s_waitcnt vmcnt(0)); tags are unique per launch, so a slot that
//     still holds an older entry never reads as valid;
//   * consumers take POSITIONS of their input queue with one atomic per wave, whether or not the entries exist yet, and poll their
//     position once per loop iteration (never in a loop of their own); a position at or beyond the producer's final count will never be
//     written: known once the producer's last workgroup has set its `done` word;
//   * everything that changes hands inside the launch pair — queue planes, validity words, counts, done words — is written with
//     write-through (agent-scope, `sc1`) stores and read with agent-scope loads: private L1s and per-XCD L2s are not refreshed by other
//     CUs' stores.
// Every wait is bounded (kRqIdleCap polls with nothing to trace: seconds): a logic error becomes an error word and RT_ERR_HIP, not a hung
// GPU. Per pixel the operations and their order are SHOOT's: the frame is bit-identical.
#pragma once
#include "rt_kernels.h"

namespace rt {

struct StreamQueue {
    QueueDev q;              // the ray planes (34 B per slot)
    unsigned long long* v;   // per slot {tag : 32, rng word : 32}: written last
    uint32_t* count;         // slots handed out so far (blocks of kOutBlock); final once `done[1]` is set
    uint32_t* done;          // [0] workgroups of the producing launch that have exited, [1] 1 = the producing launch is complete
    uint32_t tag;            // of the launch that produces this queue
    uint32_t cap;            // slots allocated: positions beyond are never written (a consumer may hold one until the producer is done)
};

#ifndef RT_WS_NOWAIT
#define RT_WS_NOWAIT 0 // experiment (only valid with RT_WS_SERIAL=1)
#endif
#ifndef RT_WS_PLAIN
#define RT_WS_PLAIN 0 // experiment (only valid with RT_WS_SERIAL=1): plain loads / stores instead of write-through ones
#endif
#if RT_WS_PLAIN
RT_DEV unsigned long long sq_load64(const unsigned long long* p) { return *p; }
RT_DEV uint32_t sq_load32(const uint32_t* p) { return *p; }
RT_DEV void sq_store64(unsigned long long* p, unsigned long long x) { *p = x; }
RT_DEV void sq_store32(uint32_t* p, uint32_t x) { *p = x; }
#else
RT_DEV unsigned long long sq_load64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RT_DEV uint32_t sq_load32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RT_DEV void sq_store64(unsigned long long* p, unsigned long long x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RT_DEV void sq_store32(uint32_t* p, uint32_t x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// an entry's planes, write-through: 16 + 8 + 8 + 4 bytes (the u16 plane is widened to a dword per slot for this schedule)
RT_DEV void sq_store_ray(const StreamQueue& Q, uint32_t slot, uint32_t id, const RayState& r) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(Q.q.org_id + slot);
    sq_store64(o, (unsigned long long)__float_as_uint(r.org.x) | ((unsigned long long)__float_as_uint(r.org.y) << 32));
    sq_store64(o + 1, (unsigned long long)__float_as_uint(r.org.z) | ((unsigned long long)id << 32));
    sq_store64(reinterpret_cast<unsigned long long*>(Q.q.a + slot),
               (unsigned long long)((uint32_t)r.dir[0] | ((uint32_t)r.dir[1] << 16)) | ((unsigned long long)((uint32_t)r.dir[2] | ((uint32_t)r.att[0] << 16)) << 32));
    sq_store64(reinterpret_cast<unsigned long long*>(Q.q.b + slot),
               (unsigned long long)((uint32_t)r.att[1] | ((uint32_t)r.att[2] << 16)) | ((unsigned long long)((uint32_t)r.rad[0] | ((uint32_t)r.rad[1] << 16)) << 32));
    sq_store32(reinterpret_cast<uint32_t*>(Q.q.c) + slot, (uint32_t)r.rad[2]);
}
RT_DEV uint32_t sq_load_ray(const StreamQueue& Q, uint32_t slot, RayState& r) {
    const unsigned long long* o = reinterpret_cast<const unsigned long long*>(Q.q.org_id + slot);
    const unsigned long long o0 = sq_load64(o), o1 = sq_load64(o + 1);
    const unsigned long long a = sq_load64(reinterpret_cast<const unsigned long long*>(Q.q.a + slot));
    const unsigned long long b = sq_load64(reinterpret_cast<const unsigned long long*>(Q.q.b + slot));
    const uint32_t c = sq_load32(reinterpret_cast<const uint32_t*>(Q.q.c) + slot);
    r.org = mk3(__uint_as_float((uint32_t)o0), __uint_as_float((uint32_t)(o0 >> 32)), __uint_as_float((uint32_t)o1));
    const uint32_t ax = (uint32_t)a, ay = (uint32_t)(a >> 32), bx = (uint32_t)b, by = (uint32_t)(b >> 32);
    r.dir[0] = (uint16_t)ax, r.dir[1] = (uint16_t)(ax >> 16), r.dir[2] = (uint16_t)ay;
    r.att[0] = (uint16_t)(ay >> 16), r.att[1] = (uint16_t)bx, r.att[2] = (uint16_t)(bx >> 16);
    r.rad[0] = (uint16_t)by, r.rad[1] = (uint16_t)(by >> 16), r.rad[2] = (uint16_t)c;
    return (uint32_t)(o1 >> 32);
}

// K3 for this schedule: camera rays of every pixel of the tile into slot = local pixel index of the bounce-0 queue, each with its validity
// word; the queue is complete when the kernel is (its consumer runs behind it on the same stream), so count and done are set here
__global__ void __launch_bounds__(256) k_ws_generate(const CameraDev* __restrict__ camp, TileDev tile, uint32_t n_local, uint32_t* __restrict__ rng, StreamQueue out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) sq_store32(out.count, n_local), sq_store32(out.done + 1, 1u);
    if (i >= n_local) return;
    const CameraDev cam = *camp;
    const int x = (int)(i % (uint32_t)cam.width), ly = (int)(i / (uint32_t)cam.width);
    const int gy = tile_global_row(tile, ly);
    uint32_t st = rng[i];
    const RayState r = gy < cam.height ? camera_ray(cam, x, gy, st) : RayState{};
    sq_store_ray(out, i, i, r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sq_store64(out.v + i, (unsigned long long)out.tag | ((unsigned long long)st << 32));
}

// One bounce: K4 (shoot_rays, src/render_wavefront.cpp:222-312) + K5 for the paths that end. `depth0` = bounces already made by the rays of `in`.
template <bool STATS>
__global__ void __launch_bounds__(kMegaBlock, kMegaWaves) k_ws_shoot(SceneDev S, TileDev tile, StreamQueue in, StreamQueue out, uint32_t* __restrict__ cursor,
                                                                    uint32_t* __restrict__ rng_buf, float4* __restrict__ accum,
                                                                    unsigned long long* __restrict__ rays_out, uint32_t depth0, uint32_t max_depth, uint32_t rr_start,
                                                                    uint32_t* __restrict__ error, unsigned long long* __restrict__ stats_out) {
    __shared__ unsigned long long out_word; // {block base : slots used} of the workgroup's current output block (SHOOT's two-level allocator)
    __shared__ uint32_t out_exited;
    if (threadIdx.x == 0) out_word = (unsigned long long)kOutBlock, out_exited = 0u; // "no block yet"
    WaveStats ws;
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t kWavesPerBlock = kMegaBlock / 64;
    RT_TRAVERSAL_LDS(kMegaBlock)
    const bool last_depth = depth0 + 1u >= max_depth; // survivors of this bounce are black: nothing is handed on
    Trav T;
    T.cur = kTravDone;
    bool live = false;    // the lane traces a ray
    bool waiting = false; // the lane holds a position of `in` whose entry is not there yet
    uint32_t pos = 0, id = 0, rng = 0;
    RayState r{};
    unsigned long long wave_rays = 0;
    uint32_t chunk_pos = 0, chunk_end = 0; // wave-uniform: positions claimed and not yet given to a lane
    bool in_done = false;                  // wave-uniform: the producing launch is complete, `final_n` is the queue's length
    uint32_t final_n = 0xFFFFFFFFu;
    uint32_t count_seen = 0, claimed_to = 0; // wave-uniform: the producer's count as last read; the end of this wave's last claim
    bool exhausted = false;                // wave-uniform: the cursor has passed the queue's end
    uint32_t idle_polls = 0;
    bool pub_pending = false; // this lane's entry of the last hand-over round is written but not yet declared valid
    uint32_t pub_slot = 0, pub_rng = 0;
    constexpr uint32_t kClaim = 256u;
    for (;;) {
        // The producer's progress — slots handed out so far, final once its done word is set — is looked at only when this wave needs it: it
        // has idle lanes and no positions left, or nothing to trace at all. (Every wave reading the count, the done word and the cursor in every
        // iteration made those three cache lines the launch's bottleneck: 1.2 ms per launch whatever the occupancy.) Positions are claimed only
        // below the count as last seen: a lane polls an entry that is being written, not one that may not exist for a long time.
        const bool want_claim = !exhausted && chunk_pos == chunk_end && __ballot(!live && !waiting) != 0ull;
        const bool nothing = __ballot(live) == 0ull;
        if (!in_done && (nothing || (want_claim && claimed_to >= count_seen))) {
            const uint32_t d = sq_load32(in.done + 1);
            count_seen = sq_load32(in.count);
            if (d != 0u) final_n = count_seen = sq_load32(in.count), in_done = true; // (read again behind the done word: final)
        }
        // idle lanes take positions
        if (!exhausted) {
            const unsigned long long idle_m = __ballot(!live && !waiting);
            const uint32_t cnt = (uint32_t)__popcll(idle_m);
            if (cnt >= 16u || cnt == 64u - (uint32_t)__popcll(__ballot(live)) ) {
                if (chunk_pos == chunk_end && cnt && (claimed_to < count_seen || in_done)) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(cursor, kClaim);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    claimed_to = base + kClaim; // (the cursor is at least here: the next claim waits until the producer's count has passed it)
                    if (in_done && base >= final_n) {
                        exhausted = true;
                    } else {
                        chunk_pos = base, chunk_end = base + kClaim;
                    }
                }
                if (chunk_pos != chunk_end) {
                    const uint32_t avail = chunk_end - chunk_pos, take = cnt < avail ? cnt : avail;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                    if (!live && !waiting && rank < take) pos = chunk_pos + rank, waiting = true;
                    chunk_pos += take;
                }
            }
        }
        // lanes that hold a position — old or just taken — look at its validity word (once per iteration, never in a loop of its own)
        if (waiting) {
            if (in_done && pos >= final_n) {
                waiting = false; // beyond the queue's end: never written
            } else if (pos < in.cap) {
                const unsigned long long vw = sq_load64(in.v + pos);
                if ((uint32_t)vw == in.tag) {
                    id = sq_load_ray(in, pos, r);
                    rng = (uint32_t)(vw >> 32);
                    waiting = false;
                    if (id != kNoRay) { // (a hole: the unused tail of a workgroup's last output block)
                        trav_begin(T, r.org, ray_dir(r), stack);
                        live = true;
                    }
                }
            }
        }
        const uint32_t n_live = (uint32_t)__popcll(__ballot(live));
        if (n_live == 0u) {
            const bool any_wait = __ballot(waiting) != 0ull;
            if (!any_wait && exhausted) break;
            if (++idle_polls > kRqIdleCap) { // seconds with nothing to trace: give up loudly rather than hang the GPU
                if (lane == 0) sq_store32(error, 4u);
                break;
            }
            if (any_wait) __builtin_amdgcn_s_sleep(8);
            else __builtin_amdgcn_s_sleep(64); // (nothing to poll but the producer's count: look rarely)
            continue;
        }
        idle_polls = 0;
        const TravSigns sg = trav_signs(T);
        for (;;) { // TRAVERSE until kMegaShadePct of the lanes that trace a ray wait for shading
            const bool done_trav = live && T.cur == kTravDone;
            if ((uint32_t)__popcll(__ballot(done_trav)) * 100u >= n_live * kMegaShadePct) break;
            if (STATS) ws.live_lane_steps += n_live * kMegaUnroll;
#pragma unroll
            for (int k = 0; k < kMegaUnroll; ++k) (void)trav_step_wave<STATS, true>(S, T, stack, top, sg, &ws);
        }
        if (STATS) ws.shade_rounds++, ws.shade_lanes += (uint32_t)__popcll(__ballot(live && T.cur == kTravDone));
        __builtin_amdgcn_s_setprio(0);
        bool hand_on = false;
        wave_rays += (unsigned long long)__popcll(__ballot(live && T.cur == kTravDone));
        if (live && T.cur == kTravDone) { // SHADE: the body of shoot_rays after rtcIntersect1 (src/render_wavefront.cpp:245-291)
            f3 res;
            const bool done = shade_bounce<true>(S, rng, r, T.best, res, &T);
            bool ends = done;
            if (done) {
                float4 acc = accum[id];
                acc.x = acc.x + clamp01(res.x), acc.y = acc.y + clamp01(res.y), acc.z = acc.z + clamp01(res.z);
                accum[id] = acc;
            } else if (last_depth) {
                ends = true; // survivors of the last depth contribute (0,0,0) (:279-280)
            } else if (rr_start && depth0 + 1u >= rr_start) {
                ends = !roulette(rng, r);
            }
            live = false;
            if (ends) rng_buf[id] = rng; // ScopedRng store: the next SAMPLE's launches read it (behind a kernel boundary)
            else hand_on = true;
        }
        { // compaction of the survivors into the next queue: src/render_wavefront.cpp:282-311
            const unsigned long long hm = __ballot(hand_on);
            if (hm != 0ull || __ballot(pub_pending) != 0ull) {
                const uint32_t n_req = (uint32_t)__popcll(hm);
                uint32_t b0 = 0, c0 = 0, b1 = 0;
                if (lane == 0 && n_req) {
                    for (;;) {
                        const unsigned long long old = atomicAdd(&out_word, (unsigned long long)n_req); // LDS
                        const uint32_t used = (uint32_t)old, base = (uint32_t)(old >> 32);
                        if (used + n_req <= kOutBlock) {
                            b0 = base + used, c0 = n_req;
                            break;
                        }
                        if (used <= kOutBlock) { // this request crosses the block's end: fill it, fetch the next block, publish it
                            b0 = base + used, c0 = kOutBlock - used;
                            b1 = atomicAdd(out.count, kOutBlock);
                            __hip_atomic_store(&out_word, ((unsigned long long)b1 << 32) | (unsigned long long)(n_req - c0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            break;
                        }
                        uint32_t polls = 0;
                        while ((uint32_t)__hip_atomic_load(&out_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > kOutBlock && ++polls < kRqIdleCap)
                            __builtin_amdgcn_s_sleep(1);
                        if (polls >= kRqIdleCap) {
                            sq_store32(error, 3u);
                            b0 = 0, c0 = n_req;
                            break;
                        }
                    }
                }
                b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0), c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
                b1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b1);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                const uint32_t slot = rank < c0 ? b0 + rank : b1 + (rank - c0);
                // The entry's planes must be retired before its validity word names them — and a write-through store takes tens of microseconds to
                // be acknowledged under load (waiting for it here, every round, was half of the launch's time). So the validity words of THIS
                // round's entries are stored in the NEXT round (or when the wave leaves), when the wait costs nothing: an entry becomes visible one
                // round late, which a consumer that runs a launch behind does not notice.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (last round's planes: long retired)
                if (pub_pending) sq_store64(out.v + pub_slot, (unsigned long long)out.tag | ((unsigned long long)pub_rng << 32));
                pub_pending = hand_on, pub_slot = slot, pub_rng = rng;
                if (hand_on) sq_store_ray(out, slot, id, r);
            }
        }
        __builtin_amdgcn_s_setprio(2);
    }
    if (lane == 0 && wave_rays) atomicAdd(rays_out, wave_rays);
    if (STATS && lane == 0) {
        atomicAdd(stats_out + 0, (unsigned long long)ws.inner_steps), atomicAdd(stats_out + 1, (unsigned long long)ws.inner_lanes);
        atomicAdd(stats_out + 2, (unsigned long long)ws.leaf_steps), atomicAdd(stats_out + 3, (unsigned long long)ws.leaf_lanes);
        atomicAdd(stats_out + 4, (unsigned long long)ws.shade_rounds), atomicAdd(stats_out + 5, (unsigned long long)ws.shade_lanes);
        atomicAdd(stats_out + 6, (unsigned long long)ws.live_lane_steps), atomicAdd(stats_out + 7, (unsigned long long)ws.top_lanes);
    }
    // the workgroup's last wave to leave marks the unused tail of the workgroup's output block as holes; the launch's last workgroup to leave
    // declares the output queue complete
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pub_pending) sq_store64(out.v + pub_slot, (unsigned long long)out.tag | ((unsigned long long)pub_rng << 32));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t order = 0;
    if (lane == 0) order = atomicAdd(&out_exited, 1u);
    order = (uint32_t)__builtin_amdgcn_readfirstlane((int)order);
    if (order + 1u != kWavesPerBlock) return;
    if (!last_depth) {
        const unsigned long long w = out_word;
        const uint32_t used = (uint32_t)w, base = (uint32_t)(w >> 32);
        for (uint32_t i = used + lane; i < kOutBlock; i += 64u) {
            RayState hole{};
            sq_store_ray(out, base + i, kNoRay, hole);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t i = used + lane; i < kOutBlock; i += 64u) sq_store64(out.v + base + i, (unsigned long long)out.tag);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    uint32_t wg = 0;
    if (lane == 0) wg = atomicAdd(out.done, 1u);
    wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)wg);
    if (wg + 1u == gridDim.x && lane == 0) sq_store32(out.done + 1, 1u);
}

} // namespace rt
