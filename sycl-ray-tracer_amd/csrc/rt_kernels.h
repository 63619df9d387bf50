// rt_kernels.h — the gfx950 kernels of the two renderers (wave64; block sizes are multiples of 64).
//
//   k_megakernel   == K1, MegakernelRenderer's single kernel (src/render_megakernel.cpp:116-168)
//   k_wf_init      == K2, WavefrontRenderer ctor init        (src/render_wavefront.cpp:62-74)
//   k_wf_generate  == K3, generate_camera_rays               (src/render_wavefront.cpp:106-124)
//   k_wf_extend    == K4a, the rtcIntersect1 half of shoot_rays (src/render_wavefront.cpp:244-273)
//   k_wf_shade     == K4b + K5, shading/compaction half of shoot_rays + merge_samples (:273-312,340-354)
//   k_wf_finish    == K4 + K5 for every bounce of a path (and, by default, K3 for every sample after a pixel's first): the
//                     WavefrontRenderer's default schedule, one launch per frame (:222-354, 396-417)
//   k_wf_resolve   == K6 convert_image_to_srgb + K7 write_image (:377-390, src/util.hpp:16-23)
#pragma once
#include <cstddef>

#include "rt_device.h"
#include "rt_launch.h"

namespace rt {

RT_DEV void queue_store(const QueueDev& q, uint32_t slot, uint32_t id, const RayState& r) {
    q.org_id[slot] = make_float4(r.org.x, r.org.y, r.org.z, __uint_as_float(id));
    q.a[slot] = make_uint2((uint32_t)r.dir[0] | ((uint32_t)r.dir[1] << 16), (uint32_t)r.dir[2] | ((uint32_t)r.att[0] << 16));
    q.b[slot] = make_uint2((uint32_t)r.att[1] | ((uint32_t)r.att[2] << 16), (uint32_t)r.rad[0] | ((uint32_t)r.rad[1] << 16));
    q.c[slot] = r.rad[2];
}
RT_DEV uint32_t queue_load(const QueueDev& q, uint32_t slot, RayState& r) {
    const float4 o = q.org_id[slot];
    const uint2 a = q.a[slot], b = q.b[slot];
    const uint16_t c = q.c[slot];
    r.org = mk3(o.x, o.y, o.z);
    r.dir[0] = (uint16_t)a.x, r.dir[1] = (uint16_t)(a.x >> 16), r.dir[2] = (uint16_t)a.y;
    r.att[0] = (uint16_t)(a.y >> 16), r.att[1] = (uint16_t)b.x, r.att[2] = (uint16_t)(b.x >> 16);
    r.rad[0] = (uint16_t)b.y, r.rad[1] = (uint16_t)(b.y >> 16), r.rad[2] = c;
    return __float_as_uint(o.w);
}

RT_DEV f3 ray_dir(const RayState& r) { return mk3(h2f(r.dir[0]), h2f(r.dir[1]), h2f(r.dir[2])); }

// the part of one bounce after the closest-hit query: unpack halves -> shade_hit -> repack (the body of
// render_pixel's loop, src/render_megakernel.cpp:34-55, and of shoot_rays, src/render_wavefront.cpp:245-291)
// FROM_TRAV (the kernels whose lanes keep a ray across shading rounds): origin and direction are READ from the traversal state — T.o
// and T.d are r.org and h2f(r.dir) exactly, trav_begin put them there — so that r.org / r.dir are written here and consumed by the
// trav_begin (or the queue store) that follows, live only inside the round: six lane registers less through the traversal loop.
// (with `tab`: the kernel's staged shading tables; without: the memory path only)
// `sink`: see shade_hit
template <bool FROM_TRAV = false, class Sink = NoSink>
RT_DEV bool shade_bounce(const SceneDev& S, uint32_t& rng, RayState& r, const Hit& h, f3& result, const Trav* T = nullptr, const ShadeTables* tab = nullptr,
                         long long* ck = nullptr, const Sink& sink = Sink{}) {
    f3 org = FROM_TRAV ? T->o : r.org;
    f3 dir = FROM_TRAV ? T->d : ray_dir(r);
    f3 att = mk3(h2f(r.att[0]), h2f(r.att[1]), h2f(r.att[2]));
    f3 rad = mk3(h2f(r.rad[0]), h2f(r.rad[1]), h2f(r.rad[2]));
    const bool done = FROM_TRAV ? shade_hit<true>(S, *tab, rng, h, org, dir, att, rad, result, ck, sink) : shade_hit<false>(S, ShadeTables{}, rng, h, org, dir, att, rad, result, nullptr, sink);
    r.org = org;
    r.dir[0] = f2h(dir.x), r.dir[1] = f2h(dir.y), r.dir[2] = f2h(dir.z);
    r.att[0] = f2h(att.x), r.att[1] = f2h(att.y), r.att[2] = f2h(att.z);
    r.rad[0] = f2h(rad.x), r.rad[1] = f2h(rad.y), r.rad[2] = f2h(rad.z);
    return done;
}

// Russian roulette on a continuing path (extension, see rt_renderer_set_russian_roulette); false = path ends
RT_DEV bool roulette(uint32_t& rng, RayState& r) {
    const float qx = h2f(r.att[0]), qy = h2f(r.att[1]), qz = h2f(r.att[2]);
    const float p = __builtin_fminf(__builtin_fmaxf(__builtin_fmaxf(qx, __builtin_fmaxf(qy, qz)), 0.05f), 1.0f);
    const float u = rng_next(rng);
    if (!(u < p)) return false;
    r.att[0] = f2h(qx / p), r.att[1] = f2h(qy / p), r.att[2] = f2h(qz / p);
    return true;
}

RT_DEV uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v; // valid in lane 0
}
// the same without overflow: a lane's ray count may be anything below 2^32 (spp x depth is bounded at 2^26 per pixel, a lane renders
// many pixels), so the halves are summed separately
RT_DEV unsigned long long wave_sum_u64(uint32_t v) {
    return (unsigned long long)wave_sum_u32(v & 0xFFFFu) + ((unsigned long long)wave_sum_u32(v >> 16) << 16);
}

// ---- K1: megakernel ---------------------------------------------------------------------------------
// One lane per pixel; a 64-thread block is one wave covering an 8x8 pixel tile. The sample loop, the
// depth loop and the BVH traversal of render_pixel are flattened into ONE wave-level state machine:
//   TRAVERSE  all lanes with a ray in flight take uniform traversal steps (trav_step_wave) until at
//             least half of the lanes that still have work are waiting for shading;
//   SHADE     the waiting lanes shade their hit, scatter, and either continue the path, start the next
//             sample (new camera ray) or retire the pixel.
// A lane therefore never idles while another lane finishes a long traversal or a long path; per-pixel
// arithmetic and its order are unchanged (bit-exact with the per-pixel loops of the reference).
// Tuning constants of the persistent traversal kernels (k_megakernel, k_wf_finish). Each was swept on the atrium and the voxel terrain;
// the sweeps are in EXPERIMENTS.md.
#ifndef RT_ROUND_PRIO
#define RT_ROUND_PRIO 0
#endif
constexpr int kMegaUnroll = 3;         // traversal steps between two checks of the loop's exit condition (1 -> 3: +1.2 %)
#ifndef RT_MEGA_SHADE_PCT
#define RT_MEGA_SHADE_PCT 75
#endif
constexpr uint32_t kMegaShadePct = RT_MEGA_SHADE_PCT; // shade when this share of the live lanes is waiting (70 / 75 / 80 / 85: 7318-7350 / 7349-7362 / 7309-7341 / 7263-7266 Mrays/s)
constexpr uint32_t kMegaRefill = 1;    // idle lanes that trigger a pixel fetch — as the STARTING value of RefillPolicy:
// A refill is an atomic round trip plus the camera-ray code, issued for the whole wave however few lanes take
// part, and a lane that waits for company is idle: with lanes falling idle at a rate L per unit of wave time, refilling at T idle lanes
// costs 64 * C * L / T + T / 2 lane-times per unit time (C: the wave time of a refill), least at T = sqrt(128 * C * L). L depends on the
// workload — a pixel lives 64 spp x 8 rays x 26 steps on the atrium, 16 spp x 1.2 rays x 3 steps on the cube — so every wave estimates
// it: pixels its lanes finished in the last 32 shading rounds, over the wave time of those rounds in units of one traversal step
// (steps taken + 5 per shading round), C = 2.5 steps: T^2 = 320 * ended / (160 + steps), T in 1 ... 16, by a chain of scalar compares
// (no clock, no float: a version with s_memtime and a float square root cost the atrium 2.5 %, the estimate from ended pixels alone
// 1-2 %). Measured with fixed thresholds 1 / 4 / 8, Mrays/s: atrium 1080p 64 spp 7241 / 7153 / 6977, 16 spp 7086 / 7035 / 6871, 4 spp
// 5580 / 6749 / 6618; Cornell box 4K 16 spp 9192 / 18106 / 18018; cube 4K 16 spp 21636 / 39292 / 50509.
struct RefillPolicy { // wave-uniform, ONE scalar register (k_megakernel is at the SGPR limit: four words of state cost it 0.9 %):
    // bits 0-4 shading rounds of the window, 5-16 pixels ended, 17-27 traversal iterations (kMegaUnroll steps each), 28-31 threshold - 1
    uint32_t w = (kMegaRefill - 1u) << 28;
    RT_DEV uint32_t threshold() const { return (w >> 28) + 1u; }
    RT_DEV void stepped() { // one iteration of the traversal loop (kMegaUnroll steps)
        if (((w >> 17) & 0x7FFu) != 0x7FFu) w += 1u << 17; // saturates: a carry would drift into the threshold field (deep trees, slow tail waves)
    }
    // after every shading round, in wave-uniform control flow; `finished`: this lane has just finished its pixel (or, where queue slots
    // are single paths, its path)
    RT_DEV void round(bool finished) {
        w += ((uint32_t)__popcll(__ballot(finished)) << 5) + 1u;
        if ((w & 31u) == 0u) { // 32 rounds
            // (the round counter's carry out of bits 0-4 has just added one to the `ended` field: taken off again)
            const uint32_t ended = ((w >> 5) & 0xFFFu) - 1u, steps = ((w >> 17) & 0x7FFu) * kMegaUnroll;
            const uint32_t num = 320u * ended, den = 160u + steps;
            const uint32_t t = num < 4u * den ? 1u : num < 9u * den ? 2u : num < 16u * den ? 3u : num < 25u * den ? 4u : num < 36u * den ? 5u
                               : num < 64u * den ? 6u : num < 100u * den ? 8u : num < 144u * den ? 10u : num < 256u * den ? 12u : 16u;
            w = (t - 1u) << 28;
        }
    }
};
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
static_assert(offsetof(MegaFrame, sl) == 56 && offsetof(SliceDev, tag0) == 8 && offsetof(SliceDev, cuts) == 16 && offsetof(SliceDev, bound) == 24 && offsetof(SliceDev, wait_cap) == 60,
              "frame_* below read MegaFrame by byte offset");
RT_DEV CameraDev frame_camera(const MegaFrame* f, int32_t width, int32_t height) { // the twelve floats as three scalar loads, here and now
    u32x4 a, b, c;
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\t"
                 "s_load_dwordx4 %1, %3, 0x10\n\t"
                 "s_load_dwordx4 %2, %3, 0x20\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c)
                 : "s"(f)); // (no memory clobber: the frame's constants are written by nobody while the kernel runs, and a clobber would pin every
                            // variable a lambda captures by reference to scratch memory)
    CameraDev cam;
    cam.center[0] = __uint_as_float(a.x), cam.center[1] = __uint_as_float(a.y), cam.center[2] = __uint_as_float(a.z);
    cam.pixel00[0] = __uint_as_float(a.w), cam.pixel00[1] = __uint_as_float(b.x), cam.pixel00[2] = __uint_as_float(b.y);
    cam.du[0] = __uint_as_float(b.z), cam.du[1] = __uint_as_float(b.w), cam.du[2] = __uint_as_float(c.x);
    cam.dv[0] = __uint_as_float(c.y), cam.dv[1] = __uint_as_float(c.z), cam.dv[2] = __uint_as_float(c.w);
    cam.width = width, cam.height = height;
    return cam;
}
struct SliceNow { // as loaded: {state pointer, tag0, shift}, cuts
    u32x4 head;
    u32x2 cuts;
};
RT_DEV SliceNow frame_slices(const MegaFrame* f) {
    SliceNow n;
    asm volatile("s_load_dwordx4 %0, %2, 0x38\n\t"
                 "s_load_dwordx2 %1, %2, 0x48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(n.head), "=&s"(n.cuts)
                 : "s"(f));
    return n;
}
RT_DEV uint32_t frame_slice_bound(const MegaFrame* f, uint32_t j) { // bound[j], j wave-uniform
    uint32_t v;
    const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(0x50u + 4u * j)); // (hipcc may keep a wave-uniform value in a vector register: "s" alone does not move it)
    asm volatile("s_load_dword %0, %1, %2\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(v)
                 : "s"(f), "s"(at));
    return v;
}
RT_DEV uint32_t frame_slice_wait_cap(const MegaFrame* f) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x74\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(v)
                 : "s"(f));
    return v;
}
// A wave of a SLICED launch that holds nothing but lanes waiting for their pixel's state (no ray traced in this round, no lane traversing): sleep ~3 us before
// the next look — a look is two 16-byte loads per lane past every cache, and at a frame's end thousands of waves may be in this state while the
// few lanes they wait for need the memory system (a hand-over a few us late costs nothing: slices take ms) — and count the looks in the wave's LDS word
// (no register through the traversal loop). After `wait_cap` looks in the wave's life — seconds of waiting: a logic error — the wave's waiting lanes take
// WHATEVER the state buffer holds (slice_wait_over) and run to their ends, so the wave leaves by its ordinary way out (an exit of its own from the
// wave's loop cost the sliced megakernel 90 more spilled registers); the caller leaves an error word when this returns true.
typedef __attribute__((address_space(3))) uint32_t lds_u32_word;
RT_DEV bool slice_wait_count(lds_u32_word* looks, const MegaFrame* f) {
    __builtin_amdgcn_s_sleep(127);
    const uint32_t n = *looks + 1u; // (every lane reads and writes the wave's word with the same value)
    *looks = n;
    return n == frame_slice_wait_cap(f) + 1u;
}
RT_DEV bool slice_wait_over(const lds_u32_word* looks, const MegaFrame* f) { return *looks > frame_slice_wait_cap(f); }
// The wave's word sits behind the three planes of per-lane colour sums, and its address is worked out from the lane's own slot in the first plane WHERE it
// is needed (the asm keeps hipcc from hoisting it out of the wave's loop): as a pointer of its own it was one more vector register through the traversal
// loop, and the pixel's column went to scratch memory for it — a scratch round trip at every path end.
RT_DEV lds_u32_word* slice_looks_word(const void __attribute__((address_space(3))) * lane_slot, const void __attribute__((address_space(3))) * planes) {
    uint32_t a = (uint32_t)(size_t)lane_slot;
    asm volatile("" : "+v"(a));
    const uint32_t base = (uint32_t)(size_t)planes;
    return (lds_u32_word*)(size_t)(base + 3u * kMegaBlock * 4u + (((a - base) >> 6) & ~3u));
}
// STATS: 0 the product; 1 TIMING ONLY (RT_KERNEL_STATS=2): cycles in traversal and in shading, shading rounds and traversal steps as four scalar
// accumulators and two clock reads per round — the instantiation whose cycles per step and per shading round are the product's within a per
// cent (the full one below holds ~20 more scalars through the loops and spills 32 dwords where the product spills 2: its rounds read 15 %
// long); 2 FULL (RT_KERNEL_STATS=1): lanes per step kind, node visits in the LDS part of the tree, the frame's timeline as well.
template <int STATS, bool SLICED>
__global__ void __launch_bounds__(kMegaBlock, kMegaWaves) k_megakernel(SceneDev S, const MegaFrame* frame, int32_t width, int32_t height, TileDev tile,
                                                    uint32_t max_depth, uint32_t spp, uint32_t seed_mode, uint32_t rr_start, float* __restrict__ out_f32,
                                                    uint8_t* __restrict__ out_u8,
                                                    unsigned long long* __restrict__ ray_counter,
                                                    unsigned long long* __restrict__ stats_out, uint32_t n_slices) {
    WaveStats ws;
    long long t_begin = 0, t_mark = 0, t_trav = 0, t_shade = 0;
    long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // RT_ROUND_STAMPS builds: cycles between the stamps of a shading round
    if (STATS) t_begin = t_mark = (long long)__builtin_readcyclecounter();
    unsigned long long wall_begin = 0, wall_exhausted = 0;
    if (STATS >= 2) wall_begin = wall_clock64(); // 100 MHz, the same counter on every CU
    const uint32_t tiles_x = (uint32_t)(width + 7) >> 3, tiles_y = (uint32_t)(tile.local_rows + 7) >> 3;
    const uint32_t n_slots = tiles_x * tiles_y * 64u; // pixel slots in 8x8-tile order (slots outside the image are skipped)
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t* pixel_cursor = reinterpret_cast<uint32_t*>(ray_counter + 1);
    int x = 0, gy = 0;  // the lane's pixel: column, global row
    uint32_t pix = 0;   // ... and its index in the tile (local row * width + x)
    uint32_t rng = 0, s = 0, depth = 0;
    unsigned long long wave_rays = 0; // rays traced by this wave (wave-uniform: a scalar pair, no lane register, no reduction at the end)
    // the pixel's colour sum lives in LDS (three planes, one slot per lane): touched once per PATH, it would otherwise hold three of the 80
    // vector registers through every traversal step of a kernel that spills
    __shared__ float color_lds[3 * kMegaBlock + (SLICED ? kMegaBlock / 64 : 0)]; // (+ SLICED: a word per wave, slice_looks_word)
    typedef __attribute__((address_space(3))) float lds_f32;
    lds_f32* const color_r = (lds_f32*)color_lds + threadIdx.x;
    lds_f32* const color_g = color_r + kMegaBlock;
    lds_f32* const color_b = color_g + kMegaBlock;
    RayState r{};
    Trav T;
    if (SLICED) *slice_looks_word(color_r, (lds_f32*)color_lds) = 0u; // per wave: rounds in which it held nothing but waiting lanes (slice_wait_count)
    RT_SHADE_LDS
    RT_TRAVERSAL_LDS(kMegaBlock)
    T.cur = kTravDone;
    // A lane is `live` when it owns a pixel slice with samples left; `depth` then holds the bounces of its current path so far — or, in a
    // SLICED launch, kPend: the lane has taken a later slice of a pixel (`s` = the sample it starts with) and waits for the state the slice before ends with. It LOOKS for that
    // state in every shading round of its wave, where a ray is started anyway (the rounds the wave's other lanes bring about; a wave with nothing
    // but waiting lanes goes from round to round, ~3 us apart). A state poll of its own beside the refill, with its own copy of the camera-ray code and its own way round the wave's
    // loop, cost more registers than the kernel has: 51 spilled, 122 -> 145 ms.
    constexpr uint32_t kPend = 0x40000000u;
    bool live = false;
    RefillPolicy refill;    // idle lanes that trigger a pixel fetch: set per wave from the pixels it finishes
    bool exhausted = false; // wave-uniform: the frame's pixel cursor has passed the last slot
    uint32_t slice_j = 0;   // wave-uniform: the slice the cursor was in at this wave's last claim
    auto state_rsrc = [&](const SliceNow& n) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)n.head.y << 32) | n.head.x), 0, (int)((uint32_t)tile.local_rows * (uint32_t)width * 32u), 0x00020000);
    };
    // (ONE frame-wide cursor. Round 1 cut the frame into a region per XCD, for L2 locality: +4 % then, -2 % ... -16 % once the kernel was bound
    // by instruction issue — all waves should advance over one front, or the frame's last pixels all lie in the slowest region. EXPERIMENTS.md)
    auto take_slot = [&](uint32_t p, uint32_t first) { // the slice of pixel slot p (8x8-tile order) that starts with sample `first` becomes this lane's; slots outside the image are skipped
        const uint32_t t = p >> 6; // (slots in row order instead of 8x8 blocks: +-0 on both scenes)
        x = (int)((t % tiles_x) * 8u + (p & 7u));
        const int ly = (int)((t / tiles_x) * 8u + ((p >> 3) & 7u));
        gy = tile_global_row(tile, ly);
        pix = (uint32_t)ly * (uint32_t)width + (uint32_t)x;
        if (x < width && ly < tile.local_rows && gy < height) {
            s = first;
            if (!SLICED || first == 0) {
                rng = pixel_seed(x, gy, width, height, seed_mode);
                *color_r = 0.0f, *color_g = 0.0f, *color_b = 0.0f;
                depth = 0;
                r = camera_ray(frame_camera(frame, width, height), x, gy, rng);
                trav_begin(T, r.org, ray_dir(r), stack);
            } else {
                depth = kPend; // the state the slice starts from comes from the lane that renders the slice before it (below)
            }
            live = true;
        }
    };
    // The chain regime: a tile with no more pixel slots than the launch has lanes is ONE generation of pixel chains — nothing to hand
    // out later, and the frame takes as long as its slowest WAVE. Every wave then takes its pixels at once, spread evenly over the tile
    // (lane l of wave w: slot l * waves + w) instead of one 8x8 block, so that all waves hold the same mix of cheap and expensive
    // pixels. 1/8 of the 1080p atrium frame, 64 spp: 32.9 -> 29.6 ms. (Spreading the slots of the dynamic regime the same way changes
    // nothing on 1/2 and 1/4 of the frame: there idle lanes refill, and the tail is a single chain.)
    {
        const uint32_t n_waves = gridDim.x * (kMegaBlock / 64u);
        if (n_slots <= n_waves * 64u && n_slices == 1u) {
            const uint32_t p = (uint32_t)lane * n_waves + blockIdx.x * (kMegaBlock / 64u) + (threadIdx.x >> 6);
            if (p < n_slots) take_slot(p, 0u);
            exhausted = true;
            if (STATS >= 2) wall_exhausted = wall_clock64();
        }
    }
    for (;;) {
        // REFILL: idle lanes take the next slots from the frame-wide cursor, one atomic per wave and round.
        // A wave is persistent: it keeps pulling pixels until the frame is exhausted, so a lane whose
        // pixel was cheap (sky) does not idle while its neighbours finish 640-ray pixels.
        if (!exhausted) {
            const unsigned long long idle_m = __ballot(!live);
            const uint32_t cnt = (uint32_t)__popcll(idle_m);
            if (cnt >= refill.threshold()) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(pixel_cursor, cnt);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                uint32_t first_here = 0, first_next = 0; // first samples of the slice the claim starts in and of the one after it
                if (SLICED) {
                    while (slice_j + 1u < n_slices && base >= (slice_j + 1u) * n_slots) slice_j++; // (the cursor only moves forward)
                    if (slice_j) first_here = frame_slice_bound(frame, slice_j - 1u);
                    first_next = frame_slice_bound(frame, slice_j);
                }
                const uint32_t slice_base = SLICED ? slice_j * n_slots : 0u, total_slots = SLICED ? n_slots * n_slices : n_slots;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                const uint32_t q = base + rank;
                if (!live && q < total_slots) {
                    const bool next = SLICED && q - slice_base >= n_slots; // a claim of at most 64 slots crosses at most one slice boundary
                    take_slot(next ? q - slice_base - n_slots : q - slice_base, next ? first_next : first_here);
                }
                if (base + cnt >= total_slots) {
                    exhausted = true;
                    if (STATS >= 2) wall_exhausted = wall_clock64();
                }
            }
        }
        const uint32_t n_live = (uint32_t)__popcll(__ballot(live));
        if (n_live == 0) {
            if (exhausted) break;
            continue;
        }
        // TRAVERSE until kMegaShadePct of the lanes that have a ray are waiting for shading. Lanes that wait for their pixel's STATE (SLICED) are not
        // among them: counted as waiting they ended the traversal phase before its first step whenever they were three quarters of a wave, and the wave's
        // traversing lanes — which lanes of other waves may be waiting for in turn — stood still until enough states had arrived from elsewhere (seen as
        // frames of seconds with two sliced frames in flight; with a lost hand-over injected, as a wave that never ends). They look for their state
        // in the rounds the other lanes bring about, and a wave that holds nothing else looks every ~3 us (below).
        if (STATS) t_mark = (long long)__builtin_readcyclecounter();
        const TravSigns sg = trav_signs(T); // every ray of this traversal phase has been started by now
        // (in the loop's test the lanes that wait for a state count as waiting, as every lane with T.cur == kTravDone does: the threshold is raised by their number instead)
        const uint32_t n_rays = SLICED ? (uint32_t)__popcll(__ballot(live && depth < kPend)) : n_live;
        const uint32_t shade_at = n_rays * kMegaShadePct + (n_live - n_rays) * 100u;
        for (;;) {
            const bool waiting = live && T.cur == kTravDone; // every live lane is traversing or waiting (for shading, or for its pixel's state)
            if ((uint32_t)__popcll(__ballot(waiting)) * 100u >= shade_at) break;
            if (STATS >= 2) ws.live_lane_steps += n_live;
            if (STATS == 1) ws.inner_steps += (uint32_t)kMegaUnroll; // (timing only: steps of either kind, counted per loop iteration)
            refill.stepped();
            (void)trav_step_wave<(STATS >= 2), true>(S, T, stack, top, sg, &ws);
#pragma unroll
            for (int k = 1; k < kMegaUnroll; ++k) { // the exit tests are checked every kMegaUnroll steps
                if (STATS >= 2) ws.live_lane_steps += n_live;
                (void)trav_step_wave<(STATS >= 2), true>(S, T, stack, top, sg, &ws);
            }
        }
        if (STATS) {
            ws.shade_rounds++;
            if (STATS >= 2) ws.shade_lanes += (uint32_t)__popcll(__ballot(live && T.cur == kTravDone));
            const long long now = (long long)__builtin_readcyclecounter();
            t_trav += now - t_mark, t_mark = now;
        }
        // SHADE the lanes whose traversal is complete
        __builtin_amdgcn_s_setprio(RT_ROUND_PRIO);
#ifdef RT_ROUND_STAMPS
        if (STATS >= 2) ck[7] = (long long)__builtin_readcyclecounter();
#endif
        bool pixel_finished = false;
        {
            const unsigned long long traced = __ballot(live && T.cur == kTravDone && depth < kPend); // every lane of the round but those that wait for a state has traced a ray
            wave_rays += (unsigned long long)__popcll(traced);
            if (SLICED && traced == 0ull && __ballot(live && depth < kPend) == 0ull && slice_wait_count(slice_looks_word(color_r, (lds_f32*)color_lds), frame) && lane == 0) atomicMax(ray_counter + 2, 1ull); // (the host: RT_ERR_HIP)
        }
        if (live && T.cur == kTravDone) {
            bool start = false; // this lane starts a path: a camera ray (src/render_megakernel.cpp:148-151)
            if (SLICED && depth >= kPend) { // the state of the pixel: four {value, tag} granules, taken when all carry this slice's tag
                const SliceNow sn = frame_slices(frame);
                const auto rsrc = state_rsrc(sn);
                const uint32_t at = pix * 32u, tag = sn.head.z + s;
                const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)at, 0, 16 /* sc1 */);
                const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(at + 16u), 0, 16 /* sc1 */);
                if ((w0.y == tag && w0.w == tag && w1.y == tag && w1.w == tag) || slice_wait_over(slice_looks_word(color_r, (lds_f32*)color_lds), frame)) {
                    *color_r = __uint_as_float(w0.x), *color_g = __uint_as_float(w0.z), *color_b = __uint_as_float(w1.x);
                    rng = w1.z;
                    start = true;
                }
            } else {
                // An ended path's colour (paths that outlive max_depth add nothing). SLICED: added where it is made, in shade_hit's two ending branches — carried
                // to their join it was kept in scratch memory (three stores per path end, three reloads each waited for on its own). The unsliced
                // instantiation has the registers for it, and spills MORE (2 -> 12 dwords) with the additions inside the branches.
                f3 res;
                const auto add = [color_r, color_g, color_b](const f3& c) { *color_r = *color_r + c.x, *color_g = *color_g + c.y, *color_b = *color_b + c.z; };
                bool done;
                if (SLICED) {
                    done = shade_bounce<true>(S, rng, r, T.best, res, &T, &tab, STATS >= 2 ? ck : nullptr, add);
                } else {
                    done = shade_bounce<true>(S, rng, r, T.best, res, &T, &tab, STATS >= 2 ? ck : nullptr);
                    if (done) add(res);
                }
#ifdef RT_TAIL_PRIO
                __builtin_amdgcn_s_setprio(RT_TAIL_PRIO);
#endif
                depth++;
                bool killed = false;
                if (rr_start && !done && depth >= rr_start && depth < max_depth) killed = !roulette(rng, r);
                if (done || killed || depth == max_depth) {
                    s++;
                    bool handed = false; // this sample was the last of its slice: the pixel goes on in another lane
                    SliceNow sn{};
                    if (SLICED) {
                        sn = frame_slices(frame);
                        const unsigned long long cuts = ((unsigned long long)sn.cuts.y << 32) | sn.cuts.x;
                        const uint32_t sh = sn.head.w;
                        handed = s < spp && (s & ((1u << sh) - 1u)) == 0u && ((cuts >> (s >> sh)) & 1ull) != 0ull;
                    }
                    if (SLICED && handed) { // its state, tagged with the sample the next slice starts with
                        live = false, pixel_finished = true;
                        const auto rsrc = state_rsrc(sn);
                        const uint32_t at = pix * 32u, tag = sn.head.z + s;
                        u32x4 w0, w1;
                        w0.x = __float_as_uint(*color_r), w0.y = tag, w0.z = __float_as_uint(*color_g), w0.w = tag;
                        w1.x = __float_as_uint(*color_b), w1.y = tag, w1.z = rng, w1.w = tag;
                        __builtin_amdgcn_raw_buffer_store_b128(w0, rsrc, (int)at, 0, 16 /* sc1 */);
                        __builtin_amdgcn_raw_buffer_store_b128(w1, rsrc, (int)(at + 16u), 0, 16 /* sc1 */);
                    } else if (s < spp) {
                        start = true;
                    } else { // pixel finished: mean, sqrt gamma, fp32 + unorm8 stores (src/render_megakernel.cpp:154-158)
                        live = false, pixel_finished = true;
                        const float n = (float)spp;
                        const f3 c = mk3(__builtin_sqrtf(*color_r / n), __builtin_sqrtf(*color_g / n), __builtin_sqrtf(*color_b / n));
                        const uint32_t o = pix;
                        if (out_f32) reinterpret_cast<float4*>(out_f32)[o] = make_float4(c.x, c.y, c.z, 1.0f);
                        if (out_u8)
                            reinterpret_cast<uint32_t*>(out_u8)[o] = (uint32_t)to_unorm8(c.x) | ((uint32_t)to_unorm8(c.y) << 8) |
                                                                     ((uint32_t)to_unorm8(c.z) << 16) | 0xff000000u;
                    }
                }
            }
            if (start) {
                depth = 0;
                int cx = x, cy = gy;
                asm volatile("" : "+v"(cx), "+v"(cy)); // (or hipcc converts the pixel's coordinates to float where the lane takes the pixel and keeps them in scratch memory until here)
                r = camera_ray(frame_camera(frame, width, height), cx, cy, rng);
            }
            if (live && depth < kPend) trav_begin(T, r.org, ray_dir(r), stack);
            RT_STAMP_AFTER((STATS >= 2 ? ck : nullptr), 5, T.ox); // sky lanes, half conversions, path ends, camera rays, trav_begin
        }
        refill.round(pixel_finished);
        if (STATS) t_shade += (long long)__builtin_readcyclecounter() - t_mark;
        __builtin_amdgcn_s_setprio(2);
    }
    if (lane == 0 && wave_rays) atomicAdd(ray_counter, wave_rays);
    if (STATS && lane == 0) {
        atomicAdd(stats_out + 0, (unsigned long long)ws.inner_steps), atomicAdd(stats_out + 1, (unsigned long long)ws.inner_lanes);
        atomicAdd(stats_out + 2, (unsigned long long)ws.leaf_steps), atomicAdd(stats_out + 3, (unsigned long long)ws.leaf_lanes);
        atomicAdd(stats_out + 4, (unsigned long long)ws.shade_rounds), atomicAdd(stats_out + 5, (unsigned long long)ws.shade_lanes);
        atomicAdd(stats_out + 6, (unsigned long long)ws.live_lane_steps), atomicAdd(stats_out + 7, (unsigned long long)ws.top_lanes);
        // wave timeline (shader clock cycles): total residency, its maximum, time in traversal and in shading
        const unsigned long long dur = (unsigned long long)((long long)__builtin_readcyclecounter() - t_begin);
        atomicAdd(stats_out + 8, dur), atomicMax(stats_out + 9, dur), atomicAdd(stats_out + 10, 1ull);
        atomicAdd(stats_out + 11, (unsigned long long)t_trav), atomicAdd(stats_out + 12, (unsigned long long)t_shade);
        // frame timeline on the chip-wide 100 MHz clock: first wave start, first wave that found the pixel queue empty, last wave end
        if (STATS >= 2) {
            atomicMin(stats_out + 13, wall_begin);
            if (wall_exhausted) atomicMin(stats_out + 14, wall_exhausted);
            atomicMax(stats_out + 15, (unsigned long long)wall_clock64());
        }
#ifdef RT_ROUND_STAMPS
        for (int i = 1; i <= 5; ++i) atomicAdd(stats_out + 15 + i, (unsigned long long)ck[i]);
#endif
    }
}

// max_depth == 0: render_pixel traces nothing (src/render_megakernel.cpp:31) -> black frame, zero rays
__global__ void __launch_bounds__(256) k_fill_black(uint32_t n_local, float* __restrict__ out_f32, uint8_t* __restrict__ out_u8) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_local) return;
    if (out_f32) reinterpret_cast<float4*>(out_f32)[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if (out_u8) reinterpret_cast<uint32_t*>(out_u8)[i] = 0xff000000u;
}

// ---- K2: seed RNG states, zero the accumulator --------------------------------------------------------
__global__ void __launch_bounds__(256) k_wf_init(const CameraDev* __restrict__ camp, TileDev tile, uint32_t seed_mode, uint32_t n_local,
                                                  uint32_t* __restrict__ rng, float4* __restrict__ accum) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_local) return;
    const CameraDev cam = *camp; // in device memory, not a kernel argument: the frame's launches replay as a hipGraph
    const int x = (int)(i % (uint32_t)cam.width), ly = (int)(i / (uint32_t)cam.width);
    const int gy = tile_global_row(tile, ly);
    rng[i] = pixel_seed(x, gy, cam.width, cam.height, seed_mode);
    accum[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---- K3: camera rays for every pixel of the tile into queue slot = local pixel index -----------------
// spread != 0 (the chain regime of small tiles: every pixel keeps its lane for all its samples, the frame takes as long as its slowest
// WAVE): the 64 slots a wave takes are pixels spread evenly over the tile instead of 64 neighbours, so that every wave holds the same
// mix of cheap and expensive pixels. 1/8 of the 1080p atrium frame, 64 spp: 31.6 -> 29.6 ms. (Where waves refill from a queue the order
// of the slots does not matter: 8x8-tile order instead of rows was measured at +-0.)
__global__ void __launch_bounds__(256) k_wf_generate(const CameraDev* __restrict__ camp, TileDev tile, uint32_t n_local,
                                                      uint32_t* __restrict__ rng, QueueDev q,
                                                      uint32_t* __restrict__ count_out, uint32_t spread, const uint32_t* __restrict__ tile_order) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) *count_out = n_local; // "*ray_buffer_length = img_size.size()" (src/render_wavefront.cpp:104)
    if (i >= n_local) return;
    const CameraDev cam = *camp;
    const uint32_t nw = n_local >> 6;
    uint32_t id = spread && nw * 64u == n_local ? (i & 63u) * nw + (i >> 6) : i; // (a bijection only for tiles of whole chunks)
    if (tile_order) { // cost-ordered second launch (k_wf_order_tiles): slots in 8x8 blocks, the most expensive block first
        const uint32_t t = tile_order[i >> 6], w = i & 63u, tiles_x = (uint32_t)cam.width >> 3;
        id = ((t / tiles_x) * 8u + (w >> 3)) * (uint32_t)cam.width + (t % tiles_x) * 8u + (w & 7u);
    }
    const int x = (int)(id % (uint32_t)cam.width), ly = (int)(id / (uint32_t)cam.width);
    const int gy = tile_global_row(tile, ly);
    uint32_t st = rng[id];
    const RayState r = gy < cam.height ? camera_ray(cam, x, gy, st) : RayState{};
    rng[id] = st;
    queue_store(q, i, id, r);
}

// ---- longest chains first ------------------------------------------------------------------------------------------------
// A frame of a few pixel generations (1/2, 1/4 of the 1080p frame on one GPU; the Cornell box at 1024x1024) ends when its last pixel
// chain does, and in queue order the last chains to START are as long as any. Longest-processing-time-first needs the costs in
// advance, and a pixel's first sample tells: the renderer traces sample 0 of every pixel in a launch of its own, which records the
// length of every path (cost, one byte per pixel); the 8x8 blocks of the tile are then sorted by the rays their 64 first samples took
// (k_wf_tile_cost + k_wf_order_tiles: one counting sort in one workgroup), and the launch for the remaining samples takes its pixels in
// that order, the most expensive blocks first — so that what starts last is cheap. A pixel's samples are still traced one after the
// other with one RNG word: the frame is unchanged.
__global__ void __launch_bounds__(64) k_wf_tile_cost(const uint8_t* __restrict__ cost, uint32_t width, uint32_t n_tiles, uint32_t* __restrict__ key) {
    const uint32_t t = blockIdx.x, w = threadIdx.x, tiles_x = width >> 3;
    if (t >= n_tiles) return;
    uint32_t v = cost[((t / tiles_x) * 8u + (w >> 3)) * width + (t % tiles_x) * 8u + (w & 7u)];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (w == 0) key[t] = v >> 2; // 0 ... 64 * 255 / 4
}
constexpr uint32_t kOrderBins = 64u * 255u / 4u + 1u;
__global__ void __launch_bounds__(1024) k_wf_order_tiles(const uint32_t* __restrict__ key, uint32_t n_tiles, uint32_t* __restrict__ order) {
    __shared__ uint32_t bin[kOrderBins];
    for (uint32_t i = threadIdx.x; i < kOrderBins; i += 1024u) bin[i] = 0;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n_tiles; t += 1024u) atomicAdd(&bin[key[t] < kOrderBins ? key[t] : kOrderBins - 1u], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { // exclusive prefix from the most expensive bin down: bin[k] = first position of its tiles
        uint32_t run = 0;
        for (uint32_t k = kOrderBins; k-- > 0;) {
            const uint32_t c = bin[k];
            bin[k] = run;
            run += c;
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n_tiles; t += 1024u) order[atomicAdd(&bin[key[t] < kOrderBins ? key[t] : kOrderBins - 1u], 1u)] = t;
}

// ---- K4 is split in two kernels: EXTEND (closest hit) and SHADE (scatter + compaction) ---------------
// The reference's shoot_rays (src/render_wavefront.cpp:222-312) does intersection, shading and
// compaction per work-item in one kernel. On a 64-wide SIMD the three kinds of work have very different
// instruction streams; keeping them in one wave-level loop caps lane utilisation near 40 % (measured on
// the megakernel: inner steps 30.8, triangle steps 23.4, shading rounds 28.1 lanes of 64). Splitting
// removes shading from the traversal loop and lets shading run dense, 64 lanes of 64.

// EXTEND: persistent waves. A wave pulls chunks of kExtendChunk queue slots from a launch-wide cursor
// (one atomic per chunk), keeps its 64 lanes fed from the chunk, and writes (t, u, v, tri) to hits[slot].
// Inside the loop only three uniform step kinds exist: inner node, one triangle, or refill
// (store finished hits + load new rays) when at least kExtendRefill lanes ask for it.
constexpr uint32_t kExtendRefill = 16;  // idle lanes that trigger a refill from the wave's chunk
template <bool STATS>
__global__ void __launch_bounds__(kExtendBlock, kExtendWaves) k_wf_extend(SceneDev S, QueueDev qin, float4* __restrict__ hits,
                                                                    const uint32_t* __restrict__ count_in,
                                                                    uint32_t* __restrict__ cursor,
                                                                    unsigned long long* __restrict__ stats_out) {
    WaveStats ws;
    const uint32_t n = *count_in;
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t kWavesPerBlock = kExtendBlock / 64;
    const uint32_t wave_id = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); // waves of a block are independent
    const uint32_t n_waves = gridDim.x * kWavesPerBlock;
    RT_TRAVERSAL_LDS(kExtendBlock)
    Trav T;
    T.cur = kTravDone;
    bool has = false;     // this lane holds a ray whose hit has not been written yet
    bool pending = false; // ray data requested last round, traversal not started yet
    float4 pend_o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint2 pend_a = make_uint2(0u, 0u);
    uint32_t slot = 0;
    // wave-uniform chunk state. The first chunk is static (chunk index = block index) so that the
    // thousands of waves of a launch do not all start with an atomic on the same word; later chunks
    // come from the cursor, offset by the static ones.
    // (With 8 waves per SIMD asked for by every launch, the static chunks already cover a 1 M-ray queue: SHOOT's guided claims — rt_frame.hip,
    // wf_shoot_tail — were tried here too and LOSE a third, 5.2 -> 3.4 Grays/s on two stream lanes: they turn static chunks into cursor claims.)
    uint32_t chunk_pos = wave_id * kExtendChunk;
    uint32_t chunk_end = chunk_pos + kExtendChunk < n ? chunk_pos + kExtendChunk : n;
    bool exhausted = chunk_pos >= n;
    if (exhausted) return; // (after the workgroup barrier of the LDS fill)
    TravSigns sg{0ull, 0ull, 0ull};
    for (;;) {
        // start the rays whose data was requested in the previous round: the loads have been in flight
        // for a whole traversal step of the other lanes instead of stalling the wave at the request
        const bool any_pending = __ballot(pending) != 0ull;
        if (pending) {
            const f3 d = mk3(h2f((uint16_t)pend_a.x), h2f((uint16_t)(pend_a.x >> 16)), h2f((uint16_t)pend_a.y));
            trav_begin(T, mk3(pend_o.x, pend_o.y, pend_o.z), d, stack);
            pending = false;
        }
        if (any_pending) sg = trav_signs(T); // rays were started: the wave's direction-sign masks, in uniform control flow
        const bool need = T.cur == kTravDone; // finished or empty lane
        const unsigned long long need_m = __ballot(need);
        const uint32_t cnt = (uint32_t)__popcll(need_m);
        if (cnt >= kExtendRefill || cnt == 64u) {
            if (STATS) ws.shade_rounds++, ws.shade_lanes += cnt;
            if (need && has) { // store the finished hit record
                hits[slot] = make_float4(T.best.t, T.best.u, T.best.v, __uint_as_float(T.best.tri));
                has = false;
            }
            if (!exhausted) {
                if (chunk_pos == chunk_end) { // next chunk: one atomic per kExtendChunk rays
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(cursor, kExtendChunk);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + n_waves * kExtendChunk;
                    if (base >= n) exhausted = true;
                    chunk_pos = base;
                    chunk_end = base + kExtendChunk < n ? base + kExtendChunk : n;
                }
                if (!exhausted) {
                    const uint32_t avail = chunk_end - chunk_pos;
                    const uint32_t take = cnt < avail ? cnt : avail;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_m, 0u));
                    if (need && rank < take) {
                        slot = chunk_pos + rank;
                        pend_o = qin.org_id[slot];
                        pend_a = qin.a[slot];
                        pending = true;
                        has = true;
                    }
                    chunk_pos += take;
                }
            }
        }
        if (__ballot(T.cur != kTravDone) == 0ull) {
            if (__ballot(pending) != 0ull) continue; // newly requested rays start next round
            if (exhausted) break;                    // every lane has stored its hit (cnt == 64 above)
            continue;
        }
        if (STATS) ws.live_lane_steps += (uint32_t)__popcll(__ballot(has));
        (void)trav_step_wave<STATS, true>(S, T, stack, top, sg, &ws);
    }
    if (STATS && lane == 0) {
        atomicAdd(stats_out + 0, (unsigned long long)ws.inner_steps), atomicAdd(stats_out + 1, (unsigned long long)ws.inner_lanes);
        atomicAdd(stats_out + 2, (unsigned long long)ws.leaf_steps), atomicAdd(stats_out + 3, (unsigned long long)ws.leaf_lanes);
        atomicAdd(stats_out + 4, (unsigned long long)ws.shade_rounds), atomicAdd(stats_out + 5, (unsigned long long)ws.shade_lanes);
        atomicAdd(stats_out + 6, (unsigned long long)ws.live_lane_steps), atomicAdd(stats_out + 7, (unsigned long long)ws.top_lanes);
    }
}

// SHADE (+K5): one thread per queue slot, dense. Reads the ray state and its hit record, runs the part of
// trace_ray after rtcIntersect1, adds terminated paths straight into the accumulator (each pixel
// terminates at most once per sample, so K5's "combined += image" needs no pass of its own) and compacts
// the survivors into the next queue: wave64 __ballot + mbcnt prefix inside each wave, wave totals
// combined through LDS, ONE global atomic per 256-thread block (the reference: one per 16 work-items).
//
// Ray re-ordering (the reference's compaction src/render_wavefront.cpp:282-311 keeps arrival order, its
// PLAN.md:18-27 lists sorting as an idea). Two independent options, template flags, both off by default:
//   REORDER  the block's survivors are written in the order of a 9-bit key = direction octant (3 bits) | 4x4x4 cell of the
//            new origin in Morton order (6 bits): a counting sort through a 512-bin LDS histogram. The 256 slots a block
//            writes are two of EXTEND's 128-slot chunks, so a chunk holds rays that start in neighbouring cells and head into
//            the same octant.
//   MATSORT  before shading, the block permutes its 256 (ray, hit) pairs by the hit's material kind (miss / diffuse /
//            metallic / dielectric / none), so that a wave executes one branch of Material::scatter instead of three.
// Neither changes any pixel: a pixel has at most one ray in flight, its accumulator and RNG word are touched by that ray only,
// and the queue order is not part of the result (tests/test_gpu_parity.py::test_wavefront_reordering_keeps_the_frame).
RT_DEV uint32_t reorder_key(const SceneDev& S, const RayState& r) {
    auto cell = [&](float p, int a) {
        const int c = (int)((p - S.cell_lo[a]) * S.cell_scale[a]);
        return (uint32_t)(c < 0 ? 0 : (c > 3 ? 3 : c));
    };
    const uint32_t cx = cell(r.org.x, 0), cy = cell(r.org.y, 1), cz = cell(r.org.z, 2);
    // 2-bit Morton interleave: x0 y0 z0 x1 y1 z1
    const uint32_t morton = (cx & 1u) | ((cy & 1u) << 1) | ((cz & 1u) << 2) | ((cx >> 1) << 3) | ((cy >> 1) << 4) | ((cz >> 1) << 5);
    const uint32_t oct = ((uint32_t)(r.dir[0] >> 15) & 1u) | (((uint32_t)(r.dir[1] >> 15) & 1u) << 1) | (((uint32_t)(r.dir[2] >> 15) & 1u) << 2); // half sign bits
    return (oct << 6) | morton;
}

// exclusive prefix sum over 256 per-thread values (4 waves); returns this thread's offset, *total = the block's sum
RT_DEV uint32_t block_scan_256(uint32_t v, uint32_t* wave_sums /* LDS, 4 */, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += o;
    }
    if (lane == 63u) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += wave_sums[w];
    *total = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    return base + incl - v;
}

template <bool REORDER, bool MATSORT>
__global__ void __launch_bounds__(256) k_wf_shade(SceneDev S, QueueDev qin, QueueDev qout, const float4* __restrict__ hits,
                                                   uint32_t* __restrict__ rng, float4* __restrict__ accum,
                                                   const uint32_t* __restrict__ count_in,
                                                   uint32_t* __restrict__ count_out, int last_depth, int rr_active) {
    __shared__ uint32_t wave_total[4];
    __shared__ uint32_t block_base;
    __shared__ uint32_t bins[REORDER ? 512 : 1];
    __shared__ uint32_t perm[MATSORT ? 256 : 1];
    __shared__ uint32_t kind_count[MATSORT ? 8 : 1];
    const uint32_t n = *count_in;
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x * 256u >= n) return; // whole block idle (uniform exit)
    if (MATSORT) {
        // material kind of this thread's own slot (0 = miss, 1 + rt type otherwise, 7 = slot beyond the queue), then a counting sort of
        // the block's slot indices by kind: thread t shades the slot at position t of the sorted order
        uint32_t kind = 7u;
        if (i < n) {
            const uint32_t tri = __float_as_uint(hits[i].w);
            kind = 0u;
            if (tri != kNoTri) {
                const uint32_t iw = S.shade[tri].instance;
                kind = 1u + (S.mats[S.packed_mat ? (iw >> kPackedInstBits) : S.inst[iw].material].type & 3u);
            }
        }
        if (threadIdx.x < 8u) kind_count[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t rank_in_kind = atomicAdd(&kind_count[kind], 1u);
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t k = 0; k < kind; ++k) before += kind_count[k];
        perm[before + rank_in_kind] = i;
        __syncthreads();
        i = perm[threadIdx.x];
    }
    bool survive = false;
    RayState r{};
    uint32_t id = 0;
    if (i < n) {
        id = queue_load(qin, i, r);
        const float4 hr = hits[i];
        Hit hit;
        hit.t = hr.x, hit.u = hr.y, hit.v = hr.z, hit.tri = __float_as_uint(hr.w);
        uint32_t st = rng[id]; // ScopedRng load (src/render_wavefront.cpp:15-32)
        f3 res;
        const bool done = shade_bounce(S, st, r, hit, res);
        if (done) {
            float4 acc = accum[id];
            acc.x = acc.x + clamp01(res.x), acc.y = acc.y + clamp01(res.y), acc.z = acc.z + clamp01(res.z);
            accum[id] = acc;
        } else {
            survive = !last_depth; // survivors of the last depth contribute (0,0,0) (:279-280)
            if (rr_active && survive) survive = roulette(st, r);
        }
        rng[id] = st;          // ScopedRng store
    }
    if (REORDER) {
        // counting sort of the block's survivors by reorder_key: histogram, exclusive scan of the 512 bins (two per thread), scatter
        bins[threadIdx.x] = 0u, bins[threadIdx.x + 256u] = 0u;
        __syncthreads();
        uint32_t key = 0, rank_in_bin = 0;
        if (survive) {
            key = reorder_key(S, r);
            rank_in_bin = atomicAdd(&bins[key], 1u);
        }
        __syncthreads();
        const uint32_t b0 = bins[2u * threadIdx.x], b1 = bins[2u * threadIdx.x + 1u];
        uint32_t total = 0;
        const uint32_t off = block_scan_256(b0 + b1, wave_total, &total);
        if (threadIdx.x == 0) block_base = total ? atomicAdd(count_out, total) : 0u;
        __syncthreads(); // every thread has read its two bins: they may be overwritten by their offsets now
        bins[2u * threadIdx.x] = off, bins[2u * threadIdx.x + 1u] = off + b0;
        __syncthreads();
        if (survive) queue_store(qout, block_base + bins[key] + rank_in_bin, id, r);
        return;
    }
    const unsigned long long mask = __ballot(survive);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (lane == 0) wave_total[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        block_base = total ? atomicAdd(count_out, total) : 0u;
    }
    __syncthreads();
    if (survive) {
        uint32_t slot = block_base + prefix;
        for (uint32_t w = 0; w < wave; ++w) slot += wave_total[w];
        queue_store(qout, slot, id, r);
    }
}

// ---- FINISH: the rest of every path of the sample in ONE launch ------------------------------------------------------------
// The reference pays one shoot_rays launch (here: EXTEND + SHADE) per bounce whatever the queue holds
// (src/render_wavefront.cpp:396-417). After the first `first_depth` bounces — wide, coherent, worth a dense SHADE and a global
// compaction each — this kernel takes the compacted queue of bounce `first_depth` and follows every ray in it to the end of its
// path: persistent waves pull 128-slot chunks of the queue exactly as EXTEND does (__ballot + mbcnt ranks, one atomic per chunk),
// a lane traverses, shades (the same shade_bounce as SHADE), and either continues with the scattered ray in registers or adds the
// path's clamped result to the accumulator, stores the pixel's RNG word and takes the next queue slot. Per pixel the operations
// and their order are exactly SHADE's (per-sample clamp, black survivors of the last depth, roulette between shading and
// compaction), so the frame is bit-identical for every first_depth (tests: test_wavefront_finish_depth_keeps_the_frame).
// What it removes: (max_depth - first_depth) x 2 dependent launches per sample with their drain tails, the hit-record round trip
// and the 68 B/ray queue write + read of every later bounce. Rays traced beyond bounce `first_depth` are counted into *extra_rays
// (the queue lengths the host sums stop at first_depth).
// n_samples > 1 (only with first_depth == 0; the renderer's default is n_samples = spp: ONE launch per frame). A pixel's samples are
// sequential (its RNG word), but pixels are independent, so the lane whose path has ended generates the SAME pixel's next camera ray
// itself — Camera::get_ray with the RNG word it holds in a register, exactly what k_wf_generate would do after reading that word back
// from memory — and goes on, for n_samples samples, before it stores the word and takes the next queue slot. A queue slot is then a
// whole pixel-chain, so a wave takes exactly as many slots as it has idle lanes (below): this is k_megakernel's loop fed from a
// queue, and as fast.
// n_samples > 1 with a FinishRequeue (REQ; RT_WF_SAMPLES_PER_LAUNCH=C with RT_WF_REQUEUE=1): breadth first instead of depth first. The lane whose path has ended
// does not keep the pixel: it appends (pixel, samples left, RNG word) to a DEVICE-WIDE dynamic queue; any lane that is idle once the
// sample-0 queue is used up takes the pixel from there and generates its next camera ray with the word it finds in the entry (what
// k_wf_generate does after reading the word back from rng_buf). All pixels thus advance through their samples together and a launch
// of C samples ends with single paths, not with pixel-chunks: one drain tail per C samples, and every idle lane of the chip can take
// any waiting pixel.
// The queue is an array of P = pixels x (C - 1) entries — exactly the number of hand-overs of a launch — used once, front to back.
//   producers: a wave stages the entries of its ended paths in LDS and appends >= 32 at a time (or whatever it has when some lane of
//              it starves): one ticket atomic on `tail`, then one 8-byte store per entry.
//   consumers: idle lanes take POSITIONS with one atomic add on `head` per wave (>= 16 at a time), whether or not the entries exist
//              yet, and then poll their position once per loop iteration — the load is issued in one iteration and looked at in the
//              next, so nobody waits for it, and a wave whose other lanes have rays keeps traversing. A position >= P will never be
//              written: the lane retires. A wave leaves when all its lanes have retired.
//   No wave ever waits for another inside a loop of its own, so there is nothing to deadlock; every position < P is written (every
//   pixel ends each of its samples) and read exactly once.
// Hand-over between CUs (MI355X: private L1 per CU, private L2 per XCD, neither refreshed by another CU's stores): every byte that
// changes hands inside the launch — the accumulator and the queue entry {pixel | (samples left - 1) << 24, RNG word} — is written with
// write-through (`sc1`) stores and read with `sc1` loads, never through a plain load. An entry is ONE 8-byte store and carries its own
// validity (all ones until written); the accumulator (one 16-byte store) is written when the path ends, the entry after the wave's
// ticket atomic has returned, and waiting for that return (s_waitcnt vmcnt(0), in-order) retires the accumulator store first.
constexpr uint32_t kRqIdleCap = 1u << 26;
constexpr uint32_t kNoRay = 0xFFFFFFFFu; // pixel id of a hole in a SHOOT-produced queue
constexpr uint32_t kRqEmpty = 0xFFFFFFFFu, kRqStage = 96u, kRqFlush = 32u, kRqClaim = 16u;
static_assert(kRqFlush + 64u <= kRqStage, "a whole wave of new entries must fit behind an unflushed stage");
RT_DEV unsigned long long rq_load64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RT_DEV void rq_store64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// LIMIT (k_wf_finish<.., false, true>, "SHOOT": rt_schedule::fused_bounce): the lane follows its ray for `bounce_limit` bounces only and
// then hands it on — survivors are COMPACTED into the next queue `qout` (__ballot + mbcnt ranks inside the wave, one atomic per wave
// and shading round) with the pixel's RNG word stored back, exactly what the reference's shoot_rays does per bounce
// (src/render_wavefront.cpp:244-312: intersect, shade, compact, one kernel). With bounce_limit = 1 this is the reference's schedule as
// ONE launch per bounce: no hit-record round trip (EXTEND -> SHADE costs 32 B per ray) and half the launches of the split pipeline.
// SLICED (k_wf_finish<.., false, false, true>; one launch per frame only): the pixel slices of k_megakernel (above) in the queue-fed loop — slot q of the
// cursor = slice q / n of queue slot q % n; a lane knows its slice by the sample it starts with; the state that changes lanes is the RNG
// word and the pixel's three clamped sums, as four {value, tag} granules in `slices->sl.state`. Within a slice the sums live in LDS (as in
// k_megakernel), NOT in the accumulator in memory: two lanes on different XCDs adding to accum[id] one after the other with plain accesses
// leave the same bytes dirty in two L2s that do not know of each other (seen: a frame off in its last bits with eight slices). The
// accumulator is written once, by the lane that finishes the pixel. The LDS for the sums is what the unsliced kernel spends on the cached
// pixel coordinates (xg_lds): a sliced launch divides for them per sample.
template <bool STATS, bool REQ, bool LIMIT = false, bool SLICED = false>
__global__ void __launch_bounds__(kMegaBlock, kMegaWaves) k_wf_finish(SceneDev S, const CameraDev* __restrict__ camp, TileDev tile, QueueDev qin,
                                                                    const uint32_t* __restrict__ count_in, uint32_t* __restrict__ cursor,
                                                                    uint32_t* __restrict__ rng_buf, float4* __restrict__ accum,
                                                                    unsigned long long* __restrict__ extra_rays, uint32_t first_depth, uint32_t max_depth,
                                                                    uint32_t n_samples, uint32_t rr_start,
                                                                    unsigned long long* __restrict__ stats_out, FinishRequeue fq, uint32_t chunk_slots, uint8_t* __restrict__ cost_out,
                                                                    QueueDev qout = QueueDev{}, uint32_t* __restrict__ count_out = nullptr, uint32_t bounce_limit = 0,
                                                                    uint32_t qflags = 0, const MegaFrame* slices = nullptr, uint32_t n_slices = 1) {
    static_assert(!(REQ && LIMIT), "the dynamic queue and the per-bounce hand-over are different schedules");
    static_assert(!(SLICED && (REQ || LIMIT)), "pixel slices belong to the one-launch schedule");
    constexpr uint32_t kPend = 0x40000000u; // SLICED, in `depth`: the lane has taken a later slice of a pixel and waits for its state (as in k_megakernel)
    // qflags & 1 (every launch of a frame whose bounces are SHOOT launches): the input queue may hold HOLES (entries whose pixel id is
    // kNoRay: the unused tail of a workgroup's last output block, below) and is therefore longer than the number of rays in it, so every
    // traced ray is counted here, on the device, instead of by the host as a queue length.
    const bool holes = (qflags & 1u) != 0u;
    // SHOOT's compaction. One device-scope atomic on the next queue's length per wave and shading round was measured at ~6 ns each,
    // serialised (46 k per launch of 2 M rays: 270 of 600 us). So output slots are handed out in two levels: a workgroup owns a BLOCK of
    // kOutBlock slots of the next queue (one device atomic per block), and its eight independent waves take their slots from the block
    // with one LDS atomic on a 64-bit word {block base : slots used}. The wave whose request crosses the block's end fills the old block,
    // fetches the next one and publishes it; waves that arrive in between wait for the new word (the owner depends on nobody). What a
    // workgroup leaves unused in its last block when its last wave exits is marked kNoRay: <= kOutBlock - 1 holes per workgroup and launch.
    __shared__ unsigned long long out_word;
    __shared__ uint32_t out_exited;
    if (LIMIT && threadIdx.x == 0) out_word = (unsigned long long)kOutBlock, out_exited = 0u; // "no block yet": used == kOutBlock (before the LDS barrier below)
    WaveStats ws;
    const uint32_t n = *count_in;
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t kWavesPerBlock = kMegaBlock / 64;
    const uint32_t wave_id = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * kWavesPerBlock;
    RT_SHADE_LDS
    RT_TRAVERSAL_LDS(kMegaBlock)
    __shared__ unsigned long long rq_stage_mem[REQ ? kWavesPerBlock * kRqStage : 1u];
    typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
    lds_u64_t* stage = (lds_u64_t*)rq_stage_mem + (threadIdx.x >> 6) * kRqStage; // REQ: this wave's staged entries
    uint32_t staged = 0, stale = 0;                                                // wave-uniform: entries staged, shading rounds since they wait
    uint32_t idle_polls = 0;                                                       // wave-uniform: consecutive iterations with nothing to trace
    const uint32_t total_entries = REQ ? n * (n_samples - 1u) : 0u;                // P
    const auto accum_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)accum, 0, (int)(n * 16u), 0x00020000); // REQ: 16-byte sc1 accesses
    // One launch per frame: a lane renders all samples of its pixel and makes every camera ray but the first itself. Its pixel's x and
    // global row were four integer divisions per SAMPLE (id % width, id / width, and the two of tile_global_row: ~100 instructions of a
    // ~600-instruction shading round that nearly always holds a lane starting a sample); they are now taken once per PIXEL and kept,
    // packed, in one LDS word per lane (images of up to 65,535 x 65,535; larger ones divide as before).
    __shared__ uint32_t xg_lds[SLICED ? 1u : kMegaBlock];
    typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
    // (the lane's word, its address worked out from the lane's stack slot where it is needed: as a pointer of its own it was kept in scratch memory,
    // and every camera ray began with a scratch round trip)
    auto xg_word = [&]() {
        uint32_t a = stack.base;
        asm volatile("" : "+v"(a));
        return (lds_u32_t*)(size_t)(a - (uint32_t)(size_t)(lds_i32*)lds_stack + (uint32_t)(size_t)(lds_u32_t*)xg_lds);
    };
    const bool xg_packed = !SLICED && camp->width < 65536 && camp->height < 65536;
    __shared__ float sum_lds[SLICED ? 3u * kMegaBlock + kMegaBlock / 64u : 1u]; // SLICED: the pixel's three sums while this lane holds it (+ a word per wave: slice_looks_word)
    typedef __attribute__((address_space(3))) float lds_f32_t;
    lds_f32_t* const sum_r = (lds_f32_t*)sum_lds + (SLICED ? threadIdx.x : 0u);
    lds_f32_t* const sum_g = sum_r + (SLICED ? kMegaBlock : 0u);
    lds_f32_t* const sum_b = sum_g + (SLICED ? kMegaBlock : 0u);
    if (SLICED) *slice_looks_word(sum_r, (lds_f32_t*)sum_lds) = 0u; // rounds in which the wave held nothing but waiting lanes (slice_wait_count)
    bool chunk_dq = false; // REQ: the wave's current chunk is a range of POSITIONS of the dynamic queue
    bool drained = false;  // REQ: the dynamic queue has no position left (wave-uniform)
    // REQ: this lane holds a position of the dynamic queue and polls it. A waiting lane has no pixel, so its position lives in `id` and
    // the two words of the polled entry land in `samples_left` and `rng` — where they belong once the entry is there — instead of in
    // registers of their own (three VGPRs that would be live across the traversal loop of a kernel that already spills).
    bool waiting = false;
    Trav T;
    T.cur = kTravDone;
    bool live = false;
    RefillPolicy refill; // idle lanes that trigger a fetch of queue slots: set per wave from the pixels (or paths) it finishes
    RayState r{};
    uint32_t id = 0, rng = 0, depth = 0;
    unsigned long long wave_rays = 0; // rays this wave traced beyond the queue entries' first (wave-uniform: a scalar pair)
    uint32_t samples_left = 0;  // samples of this lane's pixel still to render in this launch, the current one included
    bool first_counted = false; // the current path's first ray is a queue entry (counted by the host as the queue's length)
    // SHOOT, experiment (RT_WF_SHOOT_TAIL=1, bit 31 of chunk_slots): near the queue's end a wave claims a SHARE of what is left instead of a
    // whole chunk — a wave that starts a private 128-slot chunk when the cursor reaches the end is two ray generations behind the others
    const bool tail_guided = LIMIT && (chunk_slots >> 31) != 0u;
    const uint32_t static_pct = (chunk_slots >> 16) & 0x7Fu; // guided: the share of the queue handed out statically, in per cent
    chunk_slots &= 0xFFFFu;
    // the static first chunk. Guided: static_pct of the queue in equal parts, whole generations of 64 slots, at least one
    uint32_t first_slots = chunk_slots;
    // SLICED: NO static chunk — every slot comes from the cursor. A later slice of a pixel waits for the slice before it, and that is only safe if the
    // earlier slot has been taken by a wave that RUNS: a slot a wave owns by its index alone belongs to a wave that may not be resident yet (a
    // second frame in flight, any other kernel on the device), while resident waves wait for it and hold the wave slots it needs (seen: 7.8 s per
    // frame with two sliced frames in flight)
    if (SLICED) first_slots = 0;
    if (tail_guided) {
        const uint32_t per_wave = (uint32_t)(((unsigned long long)n * static_pct / 100u) / (n_waves ? n_waves : 1u));
        first_slots = per_wave < 64u ? 64u : (per_wave & ~63u);
    }
    uint32_t seen = n_waves * first_slots;                                    // the cursor as this wave last saw it (+ the static chunks)
    const uint32_t inv_waves = tail_guided ? 0x10000u / (n_waves ? n_waves : 1u) + 1u : 0u; // share = left * inv_waves >> 16 (rounded up a little)
    const uint32_t n_total = SLICED ? n * n_slices : n; // slots of the cursor: every slice of every queue slot
    uint32_t slice_j = 0;                               // SLICED, wave-uniform: the slice the wave's current chunk starts in
    uint32_t chunk_pos = wave_id * first_slots; // first chunk static, later ones from the cursor (as EXTEND)
    uint32_t chunk_end = chunk_pos + first_slots < n_total ? chunk_pos + first_slots : n_total;
    bool exhausted = chunk_pos >= n_total; // the sample-0 queue has no chunk left for this wave
    auto state_rsrc = [&](const SliceNow& sn) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)sn.head.y << 32) | sn.head.x), 0, (int)(n * 32u), 0x00020000);
    };
    auto leave = [&]() { // SHOOT: the workgroup's last wave to exit marks the unused tail of the workgroup's output block
        if (!LIMIT) return;
        uint32_t order = 0;
        if (lane == 0) order = atomicAdd(&out_exited, 1u);
        order = (uint32_t)__builtin_amdgcn_readfirstlane((int)order);
        if (order + 1u != kWavesPerBlock) return;
        const unsigned long long w = out_word; // every other wave of the workgroup has left: the word is final
        const uint32_t used = (uint32_t)w, base = (uint32_t)(w >> 32);
        for (uint32_t i = used + lane; i < kOutBlock; i += 64u) qout.org_id[base + i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kNoRay));
    };
    if (exhausted) {
        if (!REQ) {
            leave();
            return;
        }
        chunk_pos = chunk_end = 0;
    }
    auto flush = [&]() { // REQ: append the wave's staged entries to the dynamic queue
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(fq.ctl + 0, staged);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every accumulator store this wave has made is at the L2 before an entry names its pixel
        for (uint32_t i = lane; i < staged; i += 64u)
            if (t + i < fq.capacity) rq_store64(fq.entries + t + i, stage[i]);
        if (lane == 0 && t + staged > fq.capacity) __hip_atomic_store(fq.error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // host-sized: cannot happen
        staged = 0;
    };
    for (;;) {
        if (REQ && waiting && samples_left != kRqEmpty) { // last iteration's poll found the entry: k_wf_generate's body with the word in it
            const uint32_t e = samples_left;
            id = e & 0xFFFFFFu, samples_left = (e >> 24) + 1u;
            const CameraDev cam = *camp;
            const int x = (int)(id % (uint32_t)cam.width), ly = (int)(id / (uint32_t)cam.width);
            r = camera_ray(cam, x, tile_global_row(tile, ly), rng);
            depth = 0, first_counted = false;
            trav_begin(T, r.org, ray_dir(r), stack);
            live = true, waiting = false;
        }
        if (!(REQ ? drained : exhausted)) {
            const unsigned long long idle_m = __ballot(!live && !waiting);
            const uint32_t cnt = (uint32_t)__popcll(idle_m);
            if (cnt >= refill.threshold()) {
                if (chunk_pos == chunk_end) {
                    if (!exhausted) {
                        // A slot is a whole pixel-chain when the lane keeps its pixel for several samples: then a wave takes exactly as
                        // many slots as it has idle lanes (as k_megakernel does), never a private chunk that it would still be working
                        // through when the other waves have left. Single paths are short: there a chunk per atomic is right.
                        // (SHOOT with guided chunk sizes — a share of what is left, 512 ... 64 slots, to save cursor atomics — was measured slower:
                        // 58.6 -> 66.2 ms at 16 spp; a wave that starts a long private chunk near the end sets the launch's tail.)
                        uint32_t w = !REQ && n_samples > 1u ? cnt : chunk_slots;
                        if (tail_guided) {
                            const uint32_t left = n > seen ? n - seen : 0u;
                            const uint32_t share = (uint32_t)(((unsigned long long)left * inv_waves) >> 16);
                            const uint32_t least = cnt > 16u ? cnt : 16u;
                            if (share < w) w = share > least ? share : least;
                        }
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(cursor, w);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + n_waves * first_slots;
                        seen = base + w;
                        if (base >= n_total) {
                            exhausted = true;
                        } else {
                            chunk_pos = base;
                            chunk_end = base + w < n_total ? base + w : n_total;
                        }
                    }
                    if (REQ && exhausted) { // positions of the dynamic queue, whether their entries exist yet or not
                        const uint32_t w = cnt > kRqClaim ? cnt : kRqClaim;
                        uint32_t h = 0;
                        if (lane == 0) h = atomicAdd(fq.ctl + 32, w);
                        h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
                        if (h >= total_entries) {
                            drained = true; // every hand-over of this launch already has its taker
                        } else {
                            chunk_pos = h;
                            chunk_end = h + w < total_entries ? h + w : total_entries;
                            chunk_dq = true;
                        }
                    }
                }
                if (chunk_pos != chunk_end) {
                    const uint32_t avail = chunk_end - chunk_pos;
                    const uint32_t take = cnt < avail ? cnt : avail;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                    uint32_t first_here = 0, first_next = 0, slice_base = 0; // SLICED: first samples of the slice the chunk starts in and of the next, that slice's first slot
                    if (SLICED) {
                        while (slice_j + 1u < n_slices && chunk_pos >= (slice_j + 1u) * n) slice_j++; // (chunks only move forward)
                        slice_base = slice_j * n;
                        if (slice_j) first_here = frame_slice_bound(slices, slice_j - 1u);
                        first_next = frame_slice_bound(slices, slice_j);
                    }
                    if (!live && !waiting && rank < take) {
                        if (REQ && chunk_dq) {
                            id = chunk_pos + rank; // the lane's position
                            waiting = true;
                        } else {
                            uint32_t slot = chunk_pos + rank, first = 0; // `first`: the sample the lane's slice starts with
                            if (SLICED) {
                                const bool next = slot - slice_base >= n; // (a claim of at most 64 slots crosses at most one slice boundary)
                                slot = next ? slot - slice_base - n : slot - slice_base, first = next ? first_next : first_here;
                            }
                            id = queue_load(qin, slot, r);
                            if (SLICED && first != 0u) { // a later slice: the ray in the queue is sample 0's; the pixel's state comes from the lane that renders the slice before (below)
                                depth = kPend;
                                samples_left = n_samples - first, first_counted = false;
                                live = true;
                            } else if (!(holes && id == kNoRay)) { // (a hole: the slot is consumed, the lane stays idle until the next refill)
                                rng = rng_buf[id]; // ScopedRng load (src/render_wavefront.cpp:15-32): held in a register until the lane lets the pixel go
                                depth = first_depth;
                                samples_left = n_samples, first_counted = !holes;
                                if (SLICED) *sum_r = 0.0f, *sum_g = 0.0f, *sum_b = 0.0f; // (k_wf_init left the accumulator at zero)
                                if (!REQ && n_samples > 1u && xg_packed) { // the pixel's coordinates for its later camera rays: two divisions ONCE per pixel
                                    const uint32_t w = (uint32_t)camp->width;
                                    *xg_word() = (id % w) | ((uint32_t)tile_global_row(tile, (int)(id / w)) << 16);
                                }
                                trav_begin(T, r.org, ray_dir(r), stack);
                                live = true;
                            }
                        }
                    }
                    chunk_pos += take;
                }
            }
        }
        const unsigned long long wait_m = REQ ? __ballot(waiting) : 0ull;
        if (REQ && waiting) { // looked at in the next iteration
            const unsigned long long e64 = rq_load64(fq.entries + id);
            samples_left = (uint32_t)e64, rng = (uint32_t)(e64 >> 32);
        }
        const uint32_t n_live = (uint32_t)__popcll(__ballot(live));
        if (n_live == 0) {
            if (!REQ) {
                if (exhausted) break;
                continue;
            }
            if (staged) flush(); // nothing to trace: hand over what this wave still holds
            if (wait_m != 0ull) {
                if (++idle_polls > kRqIdleCap) { // seconds without an entry: give up loudly rather than hang the GPU
                    if (lane == 0) __hip_atomic_store(fq.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(8); // a wave with nothing to trace must not take issue cycles from those that have
                continue;
            }
            if (drained) break;
            continue;
        }
        idle_polls = 0;
        const TravSigns sg = trav_signs(T); // every ray of this traversal phase has been started by now
        const uint32_t n_rays = SLICED ? (uint32_t)__popcll(__ballot(live && depth < kPend)) : n_live; // lanes that wait for a state are not among them (k_megakernel)
        const uint32_t shade_at = n_rays * kMegaShadePct + (n_live - n_rays) * 100u;
        for (;;) { // TRAVERSE until kMegaShadePct of the lanes that have a ray wait for shading
            const bool done_trav = live && T.cur == kTravDone;
            if ((uint32_t)__popcll(__ballot(done_trav)) * 100u >= shade_at) break;
            if (STATS) ws.live_lane_steps += n_live * kMegaUnroll;
            refill.stepped();
            (void)trav_step_wave<STATS, true>(S, T, stack, top, sg, &ws);
#pragma unroll
            for (int k = 1; k < kMegaUnroll; ++k) (void)trav_step_wave<STATS, true>(S, T, stack, top, sg, &ws);
        }
        if (STATS) ws.shade_rounds++, ws.shade_lanes += (uint32_t)__popcll(__ballot(live && T.cur == kTravDone));
        __builtin_amdgcn_s_setprio(0);
        bool requeue = false, slot_finished = false, hand_on = false;
        // a queue entry's first ray is the queue's length, already in the host's sum (unless the queue has holes): every other ray counts here
        {
            const unsigned long long traced = __ballot(live && T.cur == kTravDone && !(SLICED && depth >= kPend));
            wave_rays += (unsigned long long)__popcll(__ballot(live && T.cur == kTravDone && !(first_counted && depth == first_depth) && !(SLICED && depth >= kPend)));
            if (SLICED && traced == 0ull && __ballot(live && depth < kPend) == 0ull && slice_wait_count(slice_looks_word(sum_r, (lds_f32_t*)sum_lds), slices) && lane == 0 && fq.error)
                __hip_atomic_store(fq.error, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the host: RT_ERR_HIP)
        }
        if (SLICED && live && T.cur == kTravDone && depth >= kPend) { // a lane that waits for its pixel's state: four {value, tag} granules, taken when all carry this slice's tag
            const SliceNow sn = frame_slices(slices);
            const auto rsrc = state_rsrc(sn);
            const uint32_t at = id * 32u, tag = sn.head.z + (n_samples - samples_left);
            const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)at, 0, 16 /* sc1 */);
            const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(at + 16u), 0, 16 /* sc1 */);
            if ((w0.y == tag && w0.w == tag && w1.y == tag && w1.w == tag) || slice_wait_over(slice_looks_word(sum_r, (lds_f32_t*)sum_lds), slices)) {
                *sum_r = __uint_as_float(w0.x), *sum_g = __uint_as_float(w0.z), *sum_b = __uint_as_float(w1.x);
                rng = w1.z;
                const CameraDev cam = *camp;
                int x, gy;
                if (xg_packed) {
                    const uint32_t xg = *xg_word();
                    x = (int)(xg & 0xFFFFu), gy = (int)(xg >> 16);
                } else {
                    x = (int)(id % (uint32_t)cam.width), gy = tile_global_row(tile, (int)(id / (uint32_t)cam.width));
                }
                r = camera_ray(cam, x, gy, rng);
                depth = 0;
                trav_begin(T, r.org, ray_dir(r), stack);
            }
        } else if (live && T.cur == kTravDone) { // SHADE: the body of shoot_rays after rtcIntersect1 (src/render_wavefront.cpp:245-291)
            f3 res;
            const bool done = shade_bounce<true>(S, rng, r, T.best, res, &T, &tab);
            depth++;
            bool ends = done;
            if (SLICED && done) { // the same additions in the same order, on the sums this lane holds in LDS
                *sum_r = *sum_r + clamp01(res.x), *sum_g = *sum_g + clamp01(res.y), *sum_b = *sum_b + clamp01(res.z);
            } else if (done) {
                float4 acc;
                if (REQ) { // the pixel's previous sample may have been added by another CU: one 16-byte sc1 load, served by the L2
                    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(accum_rsrc, (int)(id * 16u), 0, 16 /* sc1 */);
                    acc = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
                } else {
                    acc = accum[id];
                }
                acc.x = acc.x + clamp01(res.x), acc.y = acc.y + clamp01(res.y), acc.z = acc.z + clamp01(res.z);
                if (REQ) { // write-through (sc1): the pixel's next sample may be added by another CU
                    u32x4 a;
                    a.x = __float_as_uint(acc.x), a.y = __float_as_uint(acc.y), a.z = __float_as_uint(acc.z), a.w = __float_as_uint(acc.w);
                    __builtin_amdgcn_raw_buffer_store_b128(a, accum_rsrc, (int)(id * 16u), 0, 16 /* sc1 */);
                } else {
                    accum[id] = acc;
                }
            } else if (depth == max_depth) {
                ends = true; // survivors of the last depth contribute (0,0,0) (:279-280)
            } else if (rr_start && depth >= rr_start) {
                ends = !roulette(rng, r);
            }
            bool handed = false; // SLICED: this sample was the last of its slice
            if (SLICED && ends && samples_left > 1u) {
                const SliceNow sn = frame_slices(slices);
                const unsigned long long cuts = ((unsigned long long)sn.cuts.y << 32) | sn.cuts.x;
                const uint32_t sh = sn.head.w, s_next = n_samples - (samples_left - 1u); // the sample the pixel goes on with
                handed = (s_next & ((1u << sh) - 1u)) == 0u && ((cuts >> (s_next >> sh)) & 1ull) != 0ull;
                if (handed) { // the pixel's state for the lane that takes the next slice: the three sums and the RNG word, tagged with that slice's first sample
                    const auto rsrc = state_rsrc(sn);
                    const uint32_t at = id * 32u, tag = sn.head.z + s_next;
                    u32x4 w0, w1;
                    w0.x = __float_as_uint(*sum_r), w0.y = tag, w0.z = __float_as_uint(*sum_g), w0.w = tag;
                    w1.x = __float_as_uint(*sum_b), w1.y = tag, w1.z = rng, w1.w = tag;
                    __builtin_amdgcn_raw_buffer_store_b128(w0, rsrc, (int)at, 0, 16 /* sc1 */);
                    __builtin_amdgcn_raw_buffer_store_b128(w1, rsrc, (int)(at + 16u), 0, 16 /* sc1 */);
                }
            }
            if (!REQ && ends && !handed && --samples_left) { // the same pixel's next sample: k_wf_generate's body, RNG word still in the register
                const CameraDev cam = *camp;
                int x, gy;
                if (xg_packed) { // (x, global row) as the lane noted them when it took the pixel: no integer division per sample
                    const uint32_t xg = *xg_word();
                    x = (int)(xg & 0xFFFFu), gy = (int)(xg >> 16);
                } else {
                    x = (int)(id % (uint32_t)cam.width), gy = tile_global_row(tile, (int)(id / (uint32_t)cam.width));
                }
                r = camera_ray(cam, x, gy, rng);
                depth = 0, first_counted = false;
                ends = false;
            }
            if (ends && cost_out) cost_out[id] = (uint8_t)(depth < 255u ? depth : 255u); // rays of this path: what the cost ordering sorts by
            if (SLICED && ends && !handed) accum[id] = make_float4(*sum_r, *sum_g, *sum_b, 0.0f); // the pixel's last sample: the accumulator, written once
            if (ends) {
                live = false, slot_finished = true;
                if (REQ) requeue = samples_left > 1u;
                if (!requeue) rng_buf[id] = rng; // ScopedRng store (the next launch reads it; inside this launch the word travels in the entry)
            } else if (LIMIT && depth - first_depth >= bounce_limit) {
                hand_on = true, live = false, slot_finished = true; // the survivor goes to the next bounce's queue (below)
            } else {
                trav_begin(T, r.org, ray_dir(r), stack);
            }
        }
        if (LIMIT) { // compaction of the survivors into the next queue: src/render_wavefront.cpp:282-311 (there: per 16-item work-group through LDS)
            const unsigned long long hm = __ballot(hand_on);
            if (hm != 0ull) {
                const uint32_t n_req = (uint32_t)__popcll(hm);
                uint32_t b0 = 0, c0 = 0, b1 = 0; // slots [b0, b0 + c0) of the current block, the rest from b1 on (a new block)
                if (lane == 0) {
                    for (;;) {
                        const unsigned long long old = atomicAdd(&out_word, (unsigned long long)n_req); // LDS
                        const uint32_t used = (uint32_t)old, base = (uint32_t)(old >> 32);
                        if (used + n_req <= kOutBlock) { // fits
                            b0 = base + used, c0 = n_req;
                            break;
                        }
                        if (used <= kOutBlock) { // this request crosses the block's end: fill it, fetch the next block, publish it
                            b0 = base + used, c0 = kOutBlock - used;
                            b1 = atomicAdd(count_out, kOutBlock); // the one device-scope atomic per kOutBlock rays
                            __hip_atomic_store(&out_word, ((unsigned long long)b1 << 32) | (unsigned long long)(n_req - c0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            break;
                        }
                        // the block is used up and its last taker is fetching the next one: wait for the new word, then ask again. The taker depends
                        // on nobody (one device atomic, one LDS store), so this ends after ~a microsecond; the wait is bounded all the same
                        // (~seconds): a logic error becomes an error word the host turns into RT_ERR_HIP, not a wave that never leaves.
                        uint32_t polls = 0;
                        while ((uint32_t)__hip_atomic_load(&out_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > kOutBlock && ++polls < kRqIdleCap)
                            __builtin_amdgcn_s_sleep(1);
                        if (polls >= kRqIdleCap) {
                            if (fq.error) __hip_atomic_store(fq.error, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            b0 = 0, c0 = n_req; // (the frame is lost; the slots only have to be addresses inside the queue)
                            break;
                        }
                    }
                }
                b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0), c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
                b1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b1);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                if (hand_on) {
                    queue_store(qout, rank < c0 ? b0 + rank : b1 + (rank - c0), id, r);
                    rng_buf[id] = rng; // ScopedRng store: the next bounce's launch reads the word back
                }
            }
        }
        refill.round(slot_finished);
        if (REQ) {
            const unsigned long long rq_m = __ballot(requeue);
            const uint32_t n_rq = (uint32_t)__popcll(rq_m);
            if (n_rq) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(rq_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rq_m, 0u));
                if (requeue) stage[staged + rank] = (unsigned long long)(id | ((samples_left - 2u) << 24)) | ((unsigned long long)rng << 32);
                staged += n_rq;
            }
            // append when there is enough to amortise the ticket — or at once when a lane of this wave starves (work is scarce: pass it on),
            // and in any case after a few rounds, so that no pixel waits long in a busy wave's stage
            stale = staged ? stale + 1u : 0u;
            if (staged >= kRqFlush || (staged && (wait_m != 0ull || stale >= 4u))) flush(), stale = 0;
        }
        __builtin_amdgcn_s_setprio(2);
    }
    if (lane == 0 && wave_rays) atomicAdd(extra_rays, wave_rays);
    if (STATS && lane == 0) {
        atomicAdd(stats_out + 0, (unsigned long long)ws.inner_steps), atomicAdd(stats_out + 1, (unsigned long long)ws.inner_lanes);
        atomicAdd(stats_out + 2, (unsigned long long)ws.leaf_steps), atomicAdd(stats_out + 3, (unsigned long long)ws.leaf_lanes);
        atomicAdd(stats_out + 4, (unsigned long long)ws.shade_rounds), atomicAdd(stats_out + 5, (unsigned long long)ws.shade_lanes);
        atomicAdd(stats_out + 6, (unsigned long long)ws.live_lane_steps), atomicAdd(stats_out + 7, (unsigned long long)ws.top_lanes);
    }
    leave();
}

// ---- K6 + K7: mean over samples, sqrt gamma, fp32 + unorm8 outputs ------------------------------------
__global__ void __launch_bounds__(256) k_wf_resolve(uint32_t n_local, uint32_t spp, uint32_t width, uint32_t strip_rows,
                                                     uint32_t n_lanes, uint32_t lane_index, const float4* __restrict__ accum,
                                                     float* __restrict__ out_f32, uint8_t* __restrict__ out_u8) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_local) return;
    const float4 a = accum[i];
    const float n = (float)spp;
    const f3 c = mk3(__builtin_sqrtf(a.x / n), __builtin_sqrtf(a.y / n), __builtin_sqrtf(a.z / n));
    // row r of stream-lane `lane_index` (its strips are every n_lanes-th strip of the renderer's tile) -> tile row
    const uint32_t x = i % width, r = i / width;
    const uint32_t row = ((r / strip_rows) * n_lanes + lane_index) * strip_rows + r % strip_rows;
    const size_t o = (size_t)row * width + x;
    if (out_f32) reinterpret_cast<float4*>(out_f32)[o] = make_float4(c.x, c.y, c.z, 1.0f);
    if (out_u8)
        reinterpret_cast<uint32_t*>(out_u8)[o] = (uint32_t)to_unorm8(c.x) | ((uint32_t)to_unorm8(c.y) << 8) |
                                                 ((uint32_t)to_unorm8(c.z) << 16) | 0xff000000u;
}

} // namespace rt
