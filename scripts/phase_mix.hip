// phase_mix.hip — the three step kinds of the traversal kernels as separate kernels, compiled to assembly ONLY (never launched):
// scripts/phase_cost.py reads each one's VALU opcode histogram and prices it with the calibrated per-opcode issue costs
// (profiles/r02_valu_calibration.json). Weighted with the measured steps per ray (RT_KERNEL_STATS=1) this gives the DYNAMIC
// instruction mix of k_megakernel / k_wf_finish and therefore the VALU issue ceiling the PMC counters are compared with —
// the static whole-kernel mix over-weights the long, rarely executed shading code.
#include "../sycl-ray-tracer_amd/csrc/rt_kernels.h"
using namespace rt;

// one inner step per loop iteration for every lane (state kept live across iterations through the volatile-ish loop carried T)
extern "C" __global__ void __launch_bounds__(512, 6) phase_inner(SceneDev S, uint32_t n, float* out) {
    RT_TRAVERSAL_LDS(512)
    Trav T;
    trav_begin(T, mk3(out[threadIdx.x], out[threadIdx.x + 1], out[threadIdx.x + 2]), mk3(out[threadIdx.x + 3], out[threadIdx.x + 4], out[threadIdx.x + 5]), stack);
    for (uint32_t i = 0; i < n; ++i) {
        if (T.cur >= 0) trav_inner(S, T, stack, top);
        asm volatile("; --- phase_inner step boundary" ::: "memory");
        if (T.cur < 0) T.cur = (int32_t)i; // keep lanes at inner nodes: the probe prices the step, not the traversal
    }
    out[threadIdx.x] = T.best.t + (float)T.sp + (float)T.cur;
}

extern "C" __global__ void __launch_bounds__(512, 6) phase_leaf(SceneDev S, uint32_t n, float* out) {
    RT_TRAVERSAL_LDS(512)
    Trav T;
    trav_begin(T, mk3(out[threadIdx.x], out[threadIdx.x + 1], out[threadIdx.x + 2]), mk3(out[threadIdx.x + 3], out[threadIdx.x + 4], out[threadIdx.x + 5]), stack);
    for (uint32_t i = 0; i < n; ++i) {
        T.cur = ~(int32_t)(((i + threadIdx.x) << 2) | (threadIdx.x & 3u));
        trav_leaf<true>(S, T, stack);
        asm volatile("; --- phase_leaf step boundary" ::: "memory");
    }
    out[threadIdx.x] = T.best.t + T.best.u + T.best.v + (float)T.best.tri + (float)T.cur;
}

// one shading round of the megakernel: shade_bounce + the continuation logic (next bounce / next sample's camera ray / pixel store)
extern "C" __global__ void __launch_bounds__(512, 6) phase_shade(SceneDev S, CameraDev cam, uint32_t n, uint32_t spp, uint32_t max_depth, float* out, float4* img) {
    RayState r{};
    uint32_t rng = threadIdx.x * 9781u + 1u, s = 0, depth = 0;
    f3 color = mk3(0, 0, 0);
    Trav T;
    const TravStack stack{threadIdx.x * 4u, nullptr, 2048u};
    r.org = mk3(out[threadIdx.x], out[threadIdx.x + 1], out[threadIdx.x + 2]);
    r.dir[0] = (uint16_t)threadIdx.x, r.dir[1] = 0x3c00, r.dir[2] = 0xbc00;
    for (uint32_t i = 0; i < n; ++i) {
        Hit h;
        h.t = out[i], h.u = out[i + 1], h.v = out[i + 2], h.tri = (uint32_t)i * 7u + threadIdx.x;
        f3 res;
        const bool done = shade_bounce(S, rng, r, h, res);
        depth++;
        if (done || depth == max_depth) {
            if (done) color = color + res;
            s++, depth = 0;
            if (s < spp) {
                r = camera_ray(cam, (int)threadIdx.x, (int)i, rng);
            } else {
                const float nf = (float)spp;
                img[threadIdx.x] = make_float4(__builtin_sqrtf(color.x / nf), __builtin_sqrtf(color.y / nf), __builtin_sqrtf(color.z / nf), 1.0f);
                s = 0;
            }
        }
        trav_begin(T, r.org, ray_dir(r), stack);
        out[threadIdx.x + i] = T.ix + T.ox;
        asm volatile("; --- phase_shade round boundary" ::: "memory");
    }
    out[threadIdx.x] = color.x + (float)rng;
}
