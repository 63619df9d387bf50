// rt_kernel_mega2.h — k_megakernel2: K1 (src/render_megakernel.cpp:116-168) with TWO pixel contexts per lane.
//
// In k_megakernel a lane whose traversal is complete waits for its wave's shading round: measured on the bench frame, 17 of a wave's
// 58 live lanes sit in that state at any inner step (35 take the step, 8 are at a leaf). Here a lane owns up to two pixels:
//   A  the ACTIVE context: the ray the lane is traversing (the traversal state T, attenuation / radiance halves, the pixel's RNG word
//      and counters — k_megakernel's registers);
//   P  the PARKED context, 16 dwords: either a ray READY to be traversed, or the HIT of a finished traversal waiting to be shaded.
// A lane whose traversal completes EXCHANGES its contexts — the finished ray is parked as a HIT, the parked ray becomes the active one
// and starts its traversal at once — so the lane keeps taking traversal steps while its other pixel waits for the round. Shading,
// camera rays and the pixel store only ever touch P; traversal only A. Per pixel the operations and their order are k_megakernel's (one
// RNG word per pixel, samples and bounces in sequence): which lane context a pixel sits in is scheduling and changes no result.
// The frame's last generation of pixels is handed out one per lane (a lane with two unfinished pixels at the end would take twice as
// long over them: the drain of the frame is a single pixel chain, as in k_megakernel).
#pragma once
#include "rt_kernels.h"

namespace rt {

#ifndef RT_MEGA2_BLOCK
#define RT_MEGA2_BLOCK 512
#endif
#ifndef RT_MEGA2_WAVES
#define RT_MEGA2_WAVES 4
#endif
constexpr uint32_t kMega2Block = RT_MEGA2_BLOCK; // (a workgroup's waves must spread evenly over the CU's four SIMDs: 640 threads = 3 + 3 + 2 + 2 waves left room for ONE workgroup per CU)
constexpr uint32_t kMega2Waves = RT_MEGA2_WAVES;
#ifndef RT_MEGA2_XCHG
#define RT_MEGA2_XCHG 12
#endif
#ifndef RT_MEGA2_SHADE_PCT
#define RT_MEGA2_SHADE_PCT 75
#endif
constexpr uint32_t kMega2Xchg = RT_MEGA2_XCHG;          // lanes that can exchange their contexts before the wave spends an exchange phase on them
constexpr uint32_t kMega2ShadePct = RT_MEGA2_SHADE_PCT; // shade when this share of the lanes that hold a pixel has a parked hit

enum : uint32_t { kParkNone = 0, kParkReady = 1, kParkHit = 2 };

// the parked context: a ray (READY) or a ray with its closest hit (HIT)
struct Parked {
    float ox, oy, oz;            // ray origin
    uint32_t h0, h1, h2, h3, h4; // halves: dir.x | dir.y, dir.z | att.r, att.g | att.b, rad.r | rad.g, rad.b | depth << 16
    float t, u, v;               // HIT: the closest hit
    uint32_t tri;
    uint32_t rng, s;             // the pixel's RNG word and sample index
    uint32_t xg, ly;             // pixel: x | global row << 16, local row
};

RT_DEV void park_ray(Parked& p, f3 org, uint16_t dx, uint16_t dy, uint16_t dz, const RayState& r, uint32_t depth) {
    p.ox = org.x, p.oy = org.y, p.oz = org.z;
    p.h0 = (uint32_t)dx | ((uint32_t)dy << 16), p.h1 = (uint32_t)dz | ((uint32_t)r.att[0] << 16);
    p.h2 = (uint32_t)r.att[1] | ((uint32_t)r.att[2] << 16), p.h3 = (uint32_t)r.rad[0] | ((uint32_t)r.rad[1] << 16);
    p.h4 = (uint32_t)r.rad[2] | (depth << 16);
}
RT_DEV void unpark_ray(const Parked& p, RayState& r, uint32_t& depth) {
    r.org = mk3(p.ox, p.oy, p.oz);
    r.dir[0] = (uint16_t)p.h0, r.dir[1] = (uint16_t)(p.h0 >> 16), r.dir[2] = (uint16_t)p.h1;
    r.att[0] = (uint16_t)(p.h1 >> 16), r.att[1] = (uint16_t)p.h2, r.att[2] = (uint16_t)(p.h2 >> 16);
    r.rad[0] = (uint16_t)p.h3, r.rad[1] = (uint16_t)(p.h3 >> 16), r.rad[2] = (uint16_t)p.h4;
    depth = p.h4 >> 16;
}

// Eligibility (host): width, height, spp, max_depth < 65536 (packed fields) and a tile of more than two pixel generations.
template <bool STATS>
__global__ void __launch_bounds__(kMega2Block, kMega2Waves) k_megakernel2(SceneDev S, CameraDev cam, TileDev tile, uint32_t max_depth, uint32_t spp,
                                                                         uint32_t seed_mode, uint32_t rr_start, float* __restrict__ out_f32,
                                                                         uint8_t* __restrict__ out_u8, unsigned long long* __restrict__ ray_counter,
                                                                         unsigned long long* __restrict__ stats_out) {
    WaveStats ws;
    const uint32_t tiles_x = (uint32_t)(cam.width + 7) >> 3, tiles_y = (uint32_t)(tile.local_rows + 7) >> 3;
    const uint32_t n_slots = tiles_x * tiles_y * 64u; // pixel slots in 8x8-tile order (slots outside the image are skipped)
    const uint32_t n_lanes = gridDim.x * kMega2Block;
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t* pixel_cursor = reinterpret_cast<uint32_t*>(ray_counter + 1);
    unsigned long long wave_rays = 0;
    // colour sums: two slots per lane in LDS (three planes each); P owns slot p_slot, A the other one
    __shared__ float color_lds[2 * 3 * kMega2Block];
    typedef __attribute__((address_space(3))) float lds_f32;
    lds_f32* const color0 = (lds_f32*)color_lds + threadIdx.x;
    uint32_t p_slot = 0;
    RT_TRAVERSAL_LDS(kMega2Block)
    // A
    Trav T;
    T.cur = kTravDone;
    bool has_a = false;
    RayState r{};
    uint32_t rng = 0, s = 0, depth = 0, axg = 0, aly = 0;
    // P
    Parked P{};
    uint32_t p_state = kParkNone;
    RefillPolicy refill;
    bool exhausted = false;
    bool two_ok = true; // wave-uniform: a lane that already holds a pixel may take a second one (false for the frame's last generation)

    auto start_pixel = [&](uint32_t p) { // pixel slot p (8x8-tile order) becomes this lane's PARKED context, its first camera ray ready
        const uint32_t t = p >> 6;
        const int x = (int)((t % tiles_x) * 8u + (p & 7u));
        const int ly = (int)((t / tiles_x) * 8u + ((p >> 3) & 7u));
        const int gy = tile_global_row(tile, ly);
        if (x < cam.width && ly < tile.local_rows && gy < cam.height) {
            uint32_t st = pixel_seed(x, gy, cam.width, cam.height, seed_mode);
            lds_f32* c = color0 + p_slot * (3u * kMega2Block);
            c[0] = 0.0f, c[kMega2Block] = 0.0f, c[2 * kMega2Block] = 0.0f;
            const RayState cr = camera_ray(cam, x, gy, st);
            park_ray(P, cr.org, cr.dir[0], cr.dir[1], cr.dir[2], cr, 0u);
            P.rng = st, P.s = 0, P.xg = (uint32_t)x | ((uint32_t)gy << 16), P.ly = (uint32_t)ly;
            p_state = kParkReady;
        }
    };

    TravSigns sg{0ull, 0ull, 0ull};
    for (;;) {
        // REFILL: empty parked contexts take the next pixels of the frame
        if (!exhausted) {
            const unsigned long long idle_m = __ballot(p_state == kParkNone && (two_ok || !has_a));
            const uint32_t cnt = (uint32_t)__popcll(idle_m);
            if (cnt >= refill.threshold()) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(pixel_cursor, cnt);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                const uint32_t p = base + rank;
                if (p_state == kParkNone && (two_ok || !has_a) && p < n_slots) start_pixel(p);
                if (base + cnt >= n_slots) exhausted = true;
                two_ok = base + cnt + n_lanes < n_slots; // what is left is no more than one pixel per lane of the launch: the last generation
            }
        }
        // EXCHANGE: a finished (or absent) active ray makes way for the parked READY ray; a finished active ray is parked as a HIT
        {
            const bool a_done = has_a && T.cur == kTravDone;
            const bool take = (!has_a || a_done) && p_state == kParkReady;
            const bool give = a_done && p_state != kParkHit;
            const unsigned long long xm = __ballot(take || give);
            const uint32_t n_trav = (uint32_t)__popcll(__ballot(has_a && T.cur != kTravDone));
            if (xm != 0ull && ((uint32_t)__popcll(xm) >= kMega2Xchg || n_trav == 0u || (uint32_t)__popcll(xm) * 4u >= n_trav)) {
                if (take || give) {
                    Parked n = P;
                    if (give) { // A's finished ray with its hit: what shading needs (origin and direction as trav_begin got them: r.org and h2f(r.dir))
                        park_ray(n, T.o, f2h(T.d.x), f2h(T.d.y), f2h(T.d.z), r, depth);
                        n.t = T.best.t, n.u = T.best.u, n.v = T.best.v, n.tri = T.best.tri;
                        n.rng = rng, n.s = s, n.xg = axg, n.ly = aly;
                    }
                    if (take) {
                        unpark_ray(P, r, depth);
                        rng = P.rng, s = P.s, axg = P.xg, aly = P.ly;
                        trav_begin(T, r.org, ray_dir(r), stack);
                        has_a = true;
                    } else {
                        has_a = false;
                        T.cur = kTravDone;
                    }
                    P = n;
                    p_state = give ? kParkHit : kParkNone;
                    p_slot ^= 1u; // the colour slots follow their contexts
                }
                sg = trav_signs(T); // rays were started: the wave's direction-sign masks, in uniform control flow
            }
        }
        const uint32_t n_ctx = (uint32_t)__popcll(__ballot(has_a || p_state != kParkNone));
        if (n_ctx == 0u) {
            if (exhausted) break;
            continue;
        }
        // TRAVERSE until enough hits are parked, or enough lanes could exchange, or nothing is left to traverse
        for (;;) {
            const bool a_done = has_a && T.cur == kTravDone;
            const uint32_t n_trav = (uint32_t)__popcll(__ballot(has_a && !a_done));
            if (n_trav == 0u) break;
            if ((uint32_t)__popcll(__ballot(p_state == kParkHit)) * 100u >= n_ctx * kMega2ShadePct) break;
            const uint32_t n_x = (uint32_t)__popcll(__ballot(((!has_a || a_done) && p_state == kParkReady) || (a_done && p_state == kParkNone)));
            if (n_x >= kMega2Xchg || n_x * 4u >= n_trav + 4u) break;
            if (STATS) ws.live_lane_steps += n_ctx * kMegaUnroll;
            refill.stepped();
#pragma unroll
            for (int k = 0; k < kMegaUnroll; ++k) (void)trav_step_wave<STATS, true>(S, T, stack, top, sg, &ws);
        }
        // SHADE the parked hits — when enough have gathered, or when no lane can do anything else
        {
            const unsigned long long hm = __ballot(p_state == kParkHit);
            const uint32_t n_hit = (uint32_t)__popcll(hm);
            const uint32_t n_trav = (uint32_t)__popcll(__ballot(has_a && T.cur != kTravDone));
            const uint32_t n_x = (uint32_t)__popcll(__ballot(((!has_a || (has_a && T.cur == kTravDone)) && p_state == kParkReady) ||
                                                             (has_a && T.cur == kTravDone && p_state == kParkNone)));
            if (n_hit == 0u || !(n_hit * 100u >= n_ctx * kMega2ShadePct || (n_trav == 0u && n_x == 0u))) continue;
            if (STATS) ws.shade_rounds++, ws.shade_lanes += n_hit;
            __builtin_amdgcn_s_setprio(0);
            bool pixel_finished = false;
            wave_rays += (unsigned long long)n_hit; // every parked hit is one traced ray
            if (p_state == kParkHit) {
                RayState pr;
                uint32_t pdepth;
                unpark_ray(P, pr, pdepth);
                Hit h;
                h.t = P.t, h.u = P.u, h.v = P.v, h.tri = P.tri;
                uint32_t prng = P.rng, ps = P.s;
                lds_f32* c = color0 + p_slot * (3u * kMega2Block);
                f3 res;
                const bool done = shade_bounce(S, prng, pr, h, res);
                pdepth++;
                bool killed = false;
                if (rr_start && !done && pdepth >= rr_start && pdepth < max_depth) killed = !roulette(prng, pr);
                bool keep = true;
                if (done || killed || pdepth == max_depth) {
                    if (done) c[0] = c[0] + res.x, c[kMega2Block] = c[kMega2Block] + res.y, c[2 * kMega2Block] = c[2 * kMega2Block] + res.z; // paths that outlive max_depth add (0,0,0)
                    ps++;
                    pdepth = 0;
                    const int x = (int)(P.xg & 0xFFFFu), gy = (int)(P.xg >> 16);
                    if (ps < spp) {
                        pr = camera_ray(cam, x, gy, prng);
                    } else { // pixel finished: mean, sqrt gamma, fp32 + unorm8 stores (src/render_megakernel.cpp:154-158)
                        keep = false, pixel_finished = true;
                        const float n = (float)spp;
                        const f3 col = mk3(__builtin_sqrtf(c[0] / n), __builtin_sqrtf(c[kMega2Block] / n), __builtin_sqrtf(c[2 * kMega2Block] / n));
                        const size_t o = ((size_t)P.ly * (size_t)cam.width + (size_t)x);
                        if (out_f32) reinterpret_cast<float4*>(out_f32)[o] = make_float4(col.x, col.y, col.z, 1.0f);
                        if (out_u8)
                            reinterpret_cast<uint32_t*>(out_u8)[o] = (uint32_t)to_unorm8(col.x) | ((uint32_t)to_unorm8(col.y) << 8) |
                                                                     ((uint32_t)to_unorm8(col.z) << 16) | 0xff000000u;
                    }
                }
                if (keep) {
                    park_ray(P, pr.org, pr.dir[0], pr.dir[1], pr.dir[2], pr, pdepth);
                    P.rng = prng, P.s = ps;
                    p_state = kParkReady;
                } else {
                    p_state = kParkNone;
                }
            }
            refill.round(pixel_finished);
            __builtin_amdgcn_s_setprio(2);
        }
    }
    if (lane == 0 && wave_rays) atomicAdd(ray_counter, wave_rays);
    if (STATS && lane == 0) {
        atomicAdd(stats_out + 0, (unsigned long long)ws.inner_steps), atomicAdd(stats_out + 1, (unsigned long long)ws.inner_lanes);
        atomicAdd(stats_out + 2, (unsigned long long)ws.leaf_steps), atomicAdd(stats_out + 3, (unsigned long long)ws.leaf_lanes);
        atomicAdd(stats_out + 4, (unsigned long long)ws.shade_rounds), atomicAdd(stats_out + 5, (unsigned long long)ws.shade_lanes);
        atomicAdd(stats_out + 6, (unsigned long long)ws.live_lane_steps), atomicAdd(stats_out + 7, (unsigned long long)ws.top_lanes);
    }
}

} // namespace rt
