// rt_kernel_ts.h — k_megakernel_ts: the megakernel with TRAVERSAL and SHADING decoupled inside every workgroup (round 2).
//
// Why: in k_megakernel a lane whose traversal is complete waits until 80 % of its wave is waiting too, because a shading round
// costs as much as four inner steps and only pays when it is dense. Measured at an inner step (atrium, RT_STATS_LEAFWAIT build):
// 33.1 lanes step, 8.0 sit at a leaf, **16.7 wait for shading**, 6.2 hold no pixel. Shading itself runs with 48.7 of 64 lanes.
// Here a ray is no longer tied to a lane: a workgroup owns RT_TS_SLOTS pixel slots (state in global memory, L2-resident) and two
// LDS ring queues of slot numbers; seven T-waves only traverse — a lane that finishes writes its hit record, hands the slot to the
// S queue and takes the next slot from the T queue at once — and one S-wave only shades, 64 slots at a time, dense, and hands
// every slot back with its next ray (next bounce, the pixel's next sample, or a new pixel from the frame's cursor).
// Per pixel the operations and their order are those of render_pixel (src/render_megakernel.cpp:20-63, 142-158): its rays are
// still traced and shaded one after the other, only by different lanes. The frame is bit-identical to k_megakernel's.
//
// Synchronisation (one workgroup = one CU, so workgroup scope suffices: LLVM's gfx942 memory model needs no cache invalidate for
// it, only the waits the fences insert): a producer stores the slot's state, fence(release, workgroup), pushes the slot number;
// a consumer pops, fence(acquire, workgroup), loads. Queues: tail / head tickets by one LDS atomic per wave and batch, entries
// carry their own validity (kEmpty until written), `avail` counts pushed entries and is decremented before `head` is advanced, so
// a consumer only ever waits for an entry whose producer is between its ticket and its ds_write — a few cycles, never a
// dependency cycle. No wave ever blocks on a queue being empty: it does its other duty or sleeps and polls; every spin is
// bounded (kSpinCap) and trips an error word instead of hanging the GPU.
#pragma once
#include "rt_kernels.h"

namespace rt {

#ifndef RT_TS_SLOTS
#define RT_TS_SLOTS 768u // pixel slots per workgroup: 448 in T lanes + what sits in the queues and in the S-wave
#endif
#ifndef RT_TS_T_REFILL
#define RT_TS_T_REFILL 4u // idle-or-finished lanes that make a T-wave flush + refill
#endif
#ifndef RT_TS_S_MIN
#define RT_TS_S_MIN 32 // the S-wave waits (briefly) for this many slots unless the T queue is running dry
#endif
#ifndef RT_TS_T_WAVES
#define RT_TS_T_WAVES 7u // traversal waves of a workgroup
#endif
#ifndef RT_TS_S_WAVES
#define RT_TS_S_WAVES 1u // shading waves of a workgroup
#endif
#ifndef RT_TS_OCC
#define RT_TS_OCC RT_MEGA_WAVES // waves per SIMD the register allocation is held to (3 workgroups of 8 waves = 6; of 9 waves = 7 needs 72 VGPRs)
#endif
constexpr uint32_t kTsTWaves = RT_TS_T_WAVES, kTsSWaves = RT_TS_S_WAVES, kTsBlock = 64u * (kTsTWaves + kTsSWaves), kTsTLanes = kTsTWaves * 64u,
                   kTsRing = 1024u, kTsEmpty = 0xFFFFFFFFu;
constexpr uint32_t kSpinCap = 1u << 22;
static_assert(RT_TS_SLOTS <= kTsRing, "a ring must hold every slot");

struct TsSlots { // slot-major SoA in global memory; index = blockIdx.x * RT_TS_SLOTS + slot
    float4* org_px;  // org.xyz, bits(p): the pixel's index in the frame's 8x8-tile order (k_megakernel's `p`)
    uint4* state;    // halves: (dir.x dir.y), (dir.z att.r), (att.g att.b), (rad.r rad.g)
    uint2* misc;     // x: rad.b | depth << 16,  y: RNG word
    float4* color_s; // color.xyz, bits(sample index)
    float4* hit;     // t, u, v, bits(tri): written by T, read by S
};

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) int32_t lds_s32;
struct TsQueue {
    lds_u32* ring;
    lds_s32* head;
    lds_s32* tail;
    lds_s32* avail;
};

// all 64 lanes call; lanes with `flag` append `id`
RT_DEV void ts_push(const TsQueue& q, bool flag, uint32_t id) {
    const unsigned long long m = __ballot(flag);
    const int n = __popcll(m);
    if (n == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    int pos = 0;
    if (lane == 0) pos = __hip_atomic_fetch_add(q.tail, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    pos = __builtin_amdgcn_readfirstlane(pos);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (flag) __hip_atomic_store(q.ring + (((uint32_t)pos + rank) & (kTsRing - 1u)), id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) (void)__hip_atomic_fetch_add(q.avail, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// all 64 lanes call; up to `want` entries are taken; the lanes whose `rank` (0..63, distinct among the callers that want one) is below the
// returned count receive an id. `err` is set if an entry did not become valid within kSpinCap polls (never observed; keeps the GPU safe).
RT_DEV int ts_try_pop(const TsQueue& q, int want, uint32_t rank, uint32_t& id, lds_s32* err) {
    const uint32_t lane = threadIdx.x & 63u;
    int take = 0, pos = 0;
    if (lane == 0 && want > 0) {
        const int a = __hip_atomic_load(q.avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int w = a < want ? a : want;
        if (w > 0) {
            const int old = __hip_atomic_fetch_sub(q.avail, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int got = old >= w ? w : (old > 0 ? old : 0);
            if (got < w) (void)__hip_atomic_fetch_add(q.avail, w - got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // another consumer was faster
            take = got;
            if (take > 0) pos = __hip_atomic_fetch_add(q.head, take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    take = __builtin_amdgcn_readfirstlane(take);
    if (take == 0) return 0;
    pos = __builtin_amdgcn_readfirstlane(pos);
    id = kTsEmpty;
    if ((int)rank < take) {
        lds_u32* e = q.ring + (((uint32_t)pos + rank) & (kTsRing - 1u));
        uint32_t guard = 0;
        do {
            id = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } while (id == kTsEmpty && ++guard < kSpinCap);
        if (id == kTsEmpty) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(e, kTsEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return take;
}

// the frame's pixel cursor, as k_megakernel uses it: 8 regions (one per XCD first), 8x8-tile order inside; per-wave view
struct TsPixelSource {
    uint32_t* cursor; // 8 region cursors (ray_counter + 1)
    uint32_t n_slots, region_slots, tiles_x, region, regions_done;
    bool exhausted;
};
// lanes with `want` get the next pixel slots `p` of the frame (kTsEmpty when the frame has none left). All 64 lanes call.
RT_DEV uint32_t ts_take_pixels(TsPixelSource& ps, const CameraDev& cam, const TileDev& tile, bool want) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t p_out = kTsEmpty;
    for (;;) {
        const unsigned long long m = __ballot(want && p_out == kTsEmpty);
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (cnt == 0 || ps.exhausted) break;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(ps.cursor + ps.region, cnt);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t rel = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t p = ps.region * ps.region_slots + rel;
        if (want && p_out == kTsEmpty && rel < ps.region_slots && p < ps.n_slots) {
            const uint32_t t = p >> 6;
            const int x = (int)((t % ps.tiles_x) * 8u + (p & 7u)), ly = (int)((t / ps.tiles_x) * 8u + ((p >> 3) & 7u));
            if (x < cam.width && ly < tile.local_rows && tile_global_row(tile, ly) < cam.height) p_out = p; // slots outside the image are skipped
        }
        if (base + cnt >= ps.region_slots) {
            ps.region = (ps.region + 1u) & 7u;
            ps.exhausted = ++ps.regions_done >= 8u;
        }
    }
    return p_out;
}

RT_DEV void ts_store_ray(const TsSlots& sl, size_t i, const RayState& r, uint32_t p, uint32_t depth, uint32_t rng) {
    sl.org_px[i] = make_float4(r.org.x, r.org.y, r.org.z, __uint_as_float(p));
    uint4 st;
    st.x = (uint32_t)r.dir[0] | ((uint32_t)r.dir[1] << 16), st.y = (uint32_t)r.dir[2] | ((uint32_t)r.att[0] << 16);
    st.z = (uint32_t)r.att[1] | ((uint32_t)r.att[2] << 16), st.w = (uint32_t)r.rad[0] | ((uint32_t)r.rad[1] << 16);
    sl.state[i] = st;
    sl.misc[i] = make_uint2((uint32_t)r.rad[2] | (depth << 16), rng);
}

template <bool STATS>
__global__ void __launch_bounds__(kTsBlock, RT_TS_OCC) k_megakernel_ts(SceneDev S, CameraDev cam, TileDev tile, uint32_t max_depth, uint32_t spp,
                                                                         uint32_t seed_mode, uint32_t rr_start, float* __restrict__ out_f32,
                                                                         uint8_t* __restrict__ out_u8, unsigned long long* __restrict__ ray_counter,
                                                                         TsSlots sl, unsigned long long* __restrict__ stats_out) {
    __shared__ u32x4 top_all[4 * kTopNodes]; // the four planes back to back (trav_inner's fetch addresses them from one base)
    u32x4 *const top_w0 = top_all, *const top_w1 = top_all + kTopNodes, *const top_w2 = top_all + 2 * kTopNodes, *const top_ch = top_all + 3 * kTopNodes;
    __shared__ int32_t lds_stack[kLdsStack * kTsTLanes];
    __shared__ uint32_t ring_t[kTsRing], ring_s[kTsRing];
    __shared__ int32_t ctl[12]; // 0-2 T queue head / tail / avail, 3-5 S queue, 6 live slots, 7 waves still initialising, 8 error
    int32_t spill[kStackSize - kLdsStack];
    WaveStats ws;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int32_t top_count = (int32_t)S.n_nodes < kTopNodes ? (int32_t)S.n_nodes : kTopNodes;
    top_tree_fill(S, top_w0, top_w1, top_w2, top_ch, top_count);
    for (uint32_t i = threadIdx.x; i < kTsRing; i += kTsBlock) ring_t[i] = kTsEmpty, ring_s[i] = kTsEmpty;
    if (threadIdx.x < 12u) ctl[threadIdx.x] = threadIdx.x == 7u ? (int32_t)(kTsBlock / 64u) : 0;
    constexpr uint32_t kWaves = kTsBlock / 64u;
    __syncthreads();
    const TopTree top{(lds_u32x4*)top_w0, (lds_u32x4*)top_w1, (lds_u32x4*)top_w2, (lds_u32x4*)top_ch, top_count};
    const TsQueue q_t{(lds_u32*)ring_t, (lds_s32*)&ctl[0], (lds_s32*)&ctl[1], (lds_s32*)&ctl[2]};
    const TsQueue q_s{(lds_u32*)ring_s, (lds_s32*)&ctl[3], (lds_s32*)&ctl[4], (lds_s32*)&ctl[5]};
    lds_s32* live_slots = (lds_s32*)&ctl[6];
    lds_s32* init_left = (lds_s32*)&ctl[7];
    lds_s32* err = (lds_s32*)&ctl[8];
    const size_t slot_base = (size_t)blockIdx.x * RT_TS_SLOTS;
    const uint32_t tiles_x = (uint32_t)(cam.width + 7) >> 3, tiles_y = (uint32_t)(tile.local_rows + 7) >> 3;
    TsPixelSource ps;
    ps.cursor = reinterpret_cast<uint32_t*>(ray_counter + 1);
    ps.n_slots = tiles_x * tiles_y * 64u, ps.tiles_x = tiles_x;
    ps.region_slots = ((ps.n_slots + 7u) / 8u + 63u) & ~63u;
    ps.region = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; // HW_REG_XCC_ID
    ps.regions_done = 0, ps.exhausted = false;
    auto pixel_xy = [&](uint32_t p, int& x, int& ly) {
        const uint32_t t = p >> 6;
        x = (int)((t % tiles_x) * 8u + (p & 7u)), ly = (int)((t / tiles_x) * 8u + ((p >> 3) & 7u));
    };

    // ---- INIT: every wave starts its share of the workgroup's slots (a pixel each, first camera ray) and queues them for traversal ----
    {
        constexpr uint32_t per_wave = (RT_TS_SLOTS + kWaves - 1u) / kWaves;
        for (uint32_t off = 0; off < per_wave; off += 64u) {
            const uint32_t k = off + lane, slot = wave * per_wave + k;
            const bool mine = k < per_wave && slot < RT_TS_SLOTS;
            const uint32_t p = ts_take_pixels(ps, cam, tile, mine);
            const bool ok = mine && p != kTsEmpty;
            if (ok) {
                int x, ly;
                pixel_xy(p, x, ly);
                const int gy = tile_global_row(tile, ly);
                uint32_t rng = pixel_seed(x, gy, cam.width, cam.height, seed_mode);
                const RayState r = camera_ray(cam, x, gy, rng);
                ts_store_ray(sl, slot_base + slot, r, p, 0u, rng);
                sl.color_s[slot_base + slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
            }
            const int n_ok = __popcll(__ballot(ok));
            if (lane == 0 && n_ok) (void)__hip_atomic_fetch_add(live_slots, n_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            ts_push(q_t, ok, slot);
        }
        if (lane == 0) (void)__hip_atomic_fetch_sub(init_left, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    auto block_done = [&]() {
        return __hip_atomic_load(init_left, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 &&
               __hip_atomic_load(live_slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= 0;
    };
    auto failed = [&]() { return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };

    if (wave < kTsTWaves) {
        // ================= T-wave: traversal only =================
        // Nothing in this loop waits for memory it has just asked for: a finished lane's hit record is STORED in one round and its
        // slot PUBLISHED to the S queue in the next (the release fence then finds the store long complete), and a new slot's ray is
        // REQUESTED in one round and its traversal started in the next (as k_wf_extend does with its queue). A first version that
        // stored, fenced and pushed in one go stalled the whole wave ~1-2 us per hand-over: 5,500 cycles per step instead of 2,900.
        const TravStack stack{(uint32_t)(size_t)(lds_i32*)(lds_stack + threadIdx.x), spill, 4u * kTsTLanes};
        Trav T;
        T.cur = kTravDone, T.sp = stack.base;
        bool has = false;     // this lane holds a slot: traversing it, or finished and not yet stored
        bool pub = false;     // this lane stored a hit record last round: pub_slot is published this round
        bool pending = false; // this lane requested a new slot's ray last round
        uint32_t slot = 0, pub_slot = 0;
        float4 pend_o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint2 pend_d = make_uint2(0u, 0u);
        uint32_t idle_polls = 0;
        unsigned long long c_iters = 0, c_idle = 0, c_refills = 0, c_popped = 0, c_holding = 0, c_step_cycles = 0; // STATS
        const long long t_start = STATS ? (long long)__builtin_readcyclecounter() : 0;
        __builtin_amdgcn_s_setprio(2);
        for (;;) {
            if (pending) { // the ray requested last round
                trav_begin(T, mk3(pend_o.x, pend_o.y, pend_o.z), mk3(h2f((uint16_t)pend_d.x), h2f((uint16_t)(pend_d.x >> 16)), h2f((uint16_t)pend_d.y)), stack);
                pending = false, has = true;
            }
            if (__ballot(pub) != 0ull) { // publish what was stored last round
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                ts_push(q_s, pub, pub_slot);
                pub = false;
            }
            const bool finished = has && T.cur == kTravDone;
            const bool need = !has || finished;
            const unsigned long long need_m = __ballot(need);
            const uint32_t n_need = (uint32_t)__popcll(need_m);
            const bool none_traversing = __ballot(has && !finished) == 0ull;
            if (n_need >= RT_TS_T_REFILL || none_traversing) {
                if (finished) {
                    sl.hit[slot_base + slot] = make_float4(T.best.t, T.best.u, T.best.v, __uint_as_float(T.best.tri));
                    pub = true, pub_slot = slot, has = false;
                }
                uint32_t id = kTsEmpty;
                const uint32_t rank = need ? __builtin_amdgcn_mbcnt_hi((uint32_t)(need_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_m, 0u)) : 64u;
                const int got = ts_try_pop(q_t, (int)n_need, rank, id, err);
                if (STATS) c_refills++, c_popped += (unsigned long long)got;
                if (got > 0 && need && (int)rank < got && id != kTsEmpty) {
                    slot = id;
                    pend_o = sl.org_px[slot_base + slot];
                    pend_d = *reinterpret_cast<const uint2*>(sl.state + slot_base + slot);
                    pending = true;
                }
            }
            if (__ballot(has && T.cur != kTravDone) == 0ull) { // nothing to traverse right now
                if (__ballot(has || pub || pending) != 0ull) continue; // hand-overs in flight: the next round completes them
                if (block_done() || failed() || ++idle_polls > kSpinCap) break;
                if (STATS) c_idle++;
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            idle_polls = 0;
            if (STATS) c_iters++, c_holding += (unsigned long long)__popcll(__ballot(has));
            const long long t_steps = STATS ? (long long)__builtin_readcyclecounter() : 0;
            const TravSigns sg = trav_signs(T);
            (void)trav_step_wave<STATS, true>(S, T, stack, top, sg, &ws);
#pragma unroll
            for (int k = 1; k < RT_MEGA_UNROLL; ++k) (void)trav_step_wave<STATS, true>(S, T, stack, top, sg, &ws);
            if (STATS) c_step_cycles += (unsigned long long)((long long)__builtin_readcyclecounter() - t_steps);
        }
        if (STATS && lane == 0) {
            atomicAdd(stats_out + 8, c_iters), atomicAdd(stats_out + 9, c_idle), atomicAdd(stats_out + 10, c_refills), atomicAdd(stats_out + 11, c_popped);
            atomicAdd(stats_out + 6, c_step_cycles), atomicAdd(stats_out + 12, c_holding), atomicAdd(stats_out + 13, (unsigned long long)((long long)__builtin_readcyclecounter() - t_start));
        }
    } else {
        // ================= S-wave: shading only, dense =================
        uint32_t rays = 0, idle_polls = 0;
        unsigned long long c_rounds = 0, c_cycles = 0; // STATS
        __builtin_amdgcn_s_setprio(3);
        for (;;) {
            // prefer dense batches, but never let the T queue run dry for want of shading
            const int have = __hip_atomic_load(q_s.avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int t_ready = __hip_atomic_load(q_t.avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (have < RT_TS_S_MIN && have > 0 && t_ready > 64 && idle_polls < 64u) {
                ++idle_polls;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
            uint32_t slot = kTsEmpty;
            const int got = ts_try_pop(q_s, 64, lane, slot, err);
            if (got == 0) {
                if (block_done() || failed() || ++idle_polls > kSpinCap) break;
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            idle_polls = 0;
            const long long t_round = STATS ? (long long)__builtin_readcyclecounter() : 0;
            const bool active = (int)lane < got && slot != kTsEmpty;
            const size_t i = slot_base + (active ? slot : 0u);
            RayState r{};
            uint32_t p = 0, depth = 0, rng = 0, s = 0;
            f3 color = mk3(0.0f, 0.0f, 0.0f);
            bool keep = false, want_pixel = false; // keep: the slot goes back to the T queue with a ray; want_pixel: the pixel is finished
            int x = 0, ly = 0, gy = 0;
            if (active) {
                const float4 o = sl.org_px[i];
                const uint4 st = sl.state[i];
                const uint2 mi = sl.misc[i];
                const float4 cs = sl.color_s[i];
                const float4 hr = sl.hit[i];
                r.org = mk3(o.x, o.y, o.z), p = __float_as_uint(o.w);
                r.dir[0] = (uint16_t)st.x, r.dir[1] = (uint16_t)(st.x >> 16), r.dir[2] = (uint16_t)st.y;
                r.att[0] = (uint16_t)(st.y >> 16), r.att[1] = (uint16_t)st.z, r.att[2] = (uint16_t)(st.z >> 16);
                r.rad[0] = (uint16_t)st.w, r.rad[1] = (uint16_t)(st.w >> 16), r.rad[2] = (uint16_t)mi.x;
                depth = mi.x >> 16, rng = mi.y;
                color = mk3(cs.x, cs.y, cs.z), s = __float_as_uint(cs.w);
                Hit h;
                h.t = hr.x, h.u = hr.y, h.v = hr.z, h.tri = __float_as_uint(hr.w);
                pixel_xy(p, x, ly);
                gy = tile_global_row(tile, ly);
                rays++;
                f3 res;
                const bool done = shade_bounce(S, rng, r, h, res);
                depth++;
                bool killed = false;
                if (rr_start && !done && depth >= rr_start && depth < max_depth) killed = !roulette(rng, r);
                keep = true;
                if (done || killed || depth == max_depth) {
                    if (done) color = color + res; // paths that outlive max_depth add (0,0,0)
                    s++;
                    depth = 0;
                    if (s < spp) {
                        r = camera_ray(cam, x, gy, rng);
                    } else { // pixel finished: mean, sqrt gamma, fp32 + unorm8 stores (src/render_megakernel.cpp:154-158)
                        const float n = (float)spp;
                        const f3 c = mk3(__builtin_sqrtf(color.x / n), __builtin_sqrtf(color.y / n), __builtin_sqrtf(color.z / n));
                        const size_t o2 = ((size_t)ly * (size_t)cam.width + (size_t)x);
                        if (out_f32) reinterpret_cast<float4*>(out_f32)[o2] = make_float4(c.x, c.y, c.z, 1.0f);
                        if (out_u8)
                            reinterpret_cast<uint32_t*>(out_u8)[o2] = (uint32_t)to_unorm8(c.x) | ((uint32_t)to_unorm8(c.y) << 8) |
                                                                      ((uint32_t)to_unorm8(c.z) << 16) | 0xff000000u;
                        want_pixel = true;
                    }
                }
            }
            // finished pixels: the slot takes the frame's next pixel, or retires when there is none
            const uint32_t np = ts_take_pixels(ps, cam, tile, want_pixel);
            if (want_pixel) {
                if (np != kTsEmpty) {
                    p = np;
                    pixel_xy(p, x, ly);
                    gy = tile_global_row(tile, ly);
                    rng = pixel_seed(x, gy, cam.width, cam.height, seed_mode);
                    color = mk3(0.0f, 0.0f, 0.0f), s = 0, depth = 0;
                    r = camera_ray(cam, x, gy, rng);
                } else {
                    keep = false;
                }
            }
            const int n_retired = __popcll(__ballot(want_pixel && !keep));
            if (keep) {
                ts_store_ray(sl, i, r, p, depth, rng);
                sl.color_s[i] = make_float4(color.x, color.y, color.z, __uint_as_float(s));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            ts_push(q_t, keep, slot);
            if (lane == 0 && n_retired) (void)__hip_atomic_fetch_sub(live_slots, n_retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (STATS) c_rounds++, c_cycles += (unsigned long long)((long long)__builtin_readcyclecounter() - t_round);
        }
        const unsigned long long total = wave_sum_u64(rays);
        if (STATS && lane == 0) atomicAdd(stats_out + 4, c_rounds), atomicAdd(stats_out + 5, total), atomicAdd(stats_out + 14, c_cycles);
        if (lane == 0 && total) atomicAdd(ray_counter, total);
        if (lane == 0 && failed()) atomicAdd(ray_counter + 5, 1ull); // error word: the host turns it into RT_ERR_HIP
    }
    if (STATS && lane == 0 && wave < kTsTWaves) {
        atomicAdd(stats_out + 0, (unsigned long long)ws.inner_steps), atomicAdd(stats_out + 1, (unsigned long long)ws.inner_lanes);
        atomicAdd(stats_out + 2, (unsigned long long)ws.leaf_steps), atomicAdd(stats_out + 3, (unsigned long long)ws.leaf_lanes);
        atomicAdd(stats_out + 7, (unsigned long long)ws.top_lanes);
    }
}

} // namespace rt
