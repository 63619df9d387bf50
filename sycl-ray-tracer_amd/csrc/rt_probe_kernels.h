// rt_probe_kernels.h — the kernels behind rt_intersect_batch and the rt_probe_* entry points (rt_probes.hip is the one translation unit that
// includes this header).
#pragma once
#include "rt_device.h"

namespace rt {

// ---- batch closest-hit query (rt_intersect_batch) and unit probes ------------------------------------
__global__ void __launch_bounds__(256) k_intersect_batch(SceneDev S, uint32_t n, const float* __restrict__ org,
                                                          const float* __restrict__ dir, float* __restrict__ t,
                                                          float* __restrict__ u, float* __restrict__ v,
                                                          uint32_t* __restrict__ tri) {
    RT_TRAVERSAL_LDS(256)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Hit h = intersect(S, mk3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), stack, top);
    t[i] = h.t, u[i] = h.u, v[i] = h.v, tri[i] = h.tri;
}

__global__ void k_probe_xorshift(uint32_t seed, uint32_t n, float* __restrict__ out, uint32_t* __restrict__ state_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t a = seed;
    for (uint32_t i = 0; i < n; ++i) out[i] = rng_next(a);
    *state_out = a;
}

__global__ void __launch_bounds__(256) k_probe_half(uint32_t n, const float* __restrict__ in, float* __restrict__ out,
                                                     uint16_t* __restrict__ bits) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint16_t h = f2h(in[i]);
    bits[i] = h;
    out[i] = h2f(h);
}

// rcp_rn / inv_sqrt2 (rt_device.h) against the IEEE expressions they stand for, on ALL 2^32 bit patterns: bad[0] / bad[1] count the
// inputs on which they differ (NaN results compare equal). A wave tests 64 consecutive patterns, the guard ranges are multiples of
// 2^23: every wave is wholly inside or wholly outside the fast path's range.
__global__ void __launch_bounds__(256) k_probe_rounding(unsigned long long* __restrict__ bad) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nthreads = gridDim.x * 256u;
    uint32_t b0 = 0, b1 = 0;
    for (unsigned long long i = tid; i < (1ull << 32); i += nthreads) {
        const float x = __uint_as_float((uint32_t)i);
        const float g0 = rcp_rn(x), w0 = 1.0f / x;
        const float g1 = inv_sqrt2(x), w1 = 1.0f / __builtin_sqrtf(x);
        if (__float_as_uint(g0) != __float_as_uint(w0) && !(g0 != g0 && w0 != w0)) b0++;
        if (__float_as_uint(g1) != __float_as_uint(w1) && !(g1 != g1 && w1 != w1)) b1++;
    }
    if (b0) atomicAdd(bad + 0, (unsigned long long)b0);
    if (b1) atomicAdd(bad + 1, (unsigned long long)b1);
}

__global__ void __launch_bounds__(256) k_probe_scatter(SceneDev S, uint32_t material, uint32_t n,
                                                        const float* __restrict__ dir, const float* __restrict__ normal,
                                                        const float* __restrict__ uv, const uint32_t* __restrict__ seed,
                                                        uint8_t* __restrict__ ok, float* __restrict__ out_dir,
                                                        float* __restrict__ out_att, uint32_t* __restrict__ seed_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t st = seed[i];
    f3 od, oa;
    const bool r = scatter(S, S.mats[material], st, mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                           mk3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]), uv[2 * i], uv[2 * i + 1], od, oa);
    ok[i] = r ? 1 : 0;
    out_dir[3 * i] = od.x, out_dir[3 * i + 1] = od.y, out_dir[3 * i + 2] = od.z;
    out_att[3 * i] = oa.x, out_att[3 * i + 1] = oa.y, out_att[3 * i + 2] = oa.z;
    seed_out[i] = st;
}


} // namespace rt
