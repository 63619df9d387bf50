// fast_math_verify.hip — exhaustive check (all 2^32 bit patterns) of short fma sequences against the correctly rounded
// IEEE results the arithmetic contract demands:  inv_sqrt2(x) = RN(1 / RN(sqrt(x)))  and  rcp(x) = RN(1 / x).
// Prints, per candidate, the number of mismatching inputs inside a guard range and the exponent range that is free of mismatches.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero scripts/fast_math_verify.hip -o /tmp/fmv && /tmp/fmv
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define DEV __device__ __forceinline__
DEV float ref_inv_sqrt2(float x) { return 1.0f / __builtin_sqrtf(x); }
DEV float ref_rcp(float x) { return 1.0f / x; }
DEV float ref_sqrt(float x) { return __builtin_sqrtf(x); }

// Goldschmidt square root from v_rsq_f32 with one residual correction
DEV void sqrt_core(float x, float& s, float& h) {
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r;
    h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    const float d = __builtin_fmaf(-g, g, x);
    s = __builtin_fmaf(d, h, g);
}
DEV float cand_sqrt(float x) {
    float s, h;
    sqrt_core(x, s, h);
    return s;
}
// A: reciprocal of s seeded with 2h (~1/sqrt(x)), two Newton steps
DEV float cand_isq_A(float x) {
    float s, h;
    sqrt_core(x, s, h);
    float y = h + h;
    float e = __builtin_fmaf(-s, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-s, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
// B: reciprocal of s seeded with v_rcp_f32, two Newton steps
DEV float cand_isq_B(float x) {
    float s, h;
    sqrt_core(x, s, h);
    float y = __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-s, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
// C: seeded with 2h, one Newton step
DEV float cand_isq_C(float x) {
    float s, h;
    sqrt_core(x, s, h);
    float y = h + h;
    const float e = __builtin_fmaf(-s, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
// D: seeded with 2h, residual form of the last step: q = y + y * (1 - s y) with the residual of the REFINED y
DEV float cand_isq_D(float x) {
    float s, h;
    sqrt_core(x, s, h);
    float y = h + h;
    float e = __builtin_fmaf(-s, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-s, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-s, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
DEV float cand_rcp_1(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
DEV float cand_rcp_2(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}

constexpr int kCands = 7;
__global__ void k_verify(unsigned long long* bad_by_exp /* [kCands][256] */) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    for (unsigned long long i = tid; i < (1ull << 32); i += nthreads) {
        const uint32_t b = (uint32_t)i;
        const float x = __uint_as_float(b);
        const uint32_t ex = (b >> 23) & 0xffu;
        const float got[kCands] = {cand_sqrt(x), cand_isq_A(x), cand_isq_B(x), cand_isq_C(x), cand_isq_D(x), cand_rcp_1(x), cand_rcp_2(x)};
        const float want[kCands] = {ref_sqrt(x), ref_inv_sqrt2(x), ref_inv_sqrt2(x), ref_inv_sqrt2(x), ref_inv_sqrt2(x), ref_rcp(x), ref_rcp(x)};
        for (int c = 0; c < kCands; ++c) {
            const bool sign_matters = c >= 5; // the square-root family is checked on positive inputs only
            if (!sign_matters && (b >> 31)) continue;
            if (__float_as_uint(got[c]) != __float_as_uint(want[c]) && !(got[c] != got[c] && want[c] != want[c])) atomicAdd(&bad_by_exp[c * 256 + ex], 1ull);
        }
    }
}

int main() {
    unsigned long long* d;
    hipMalloc((void**)&d, kCands * 256 * sizeof(unsigned long long));
    hipMemset(d, 0, kCands * 256 * sizeof(unsigned long long));
    hipLaunchKernelGGL(k_verify, dim3(256 * 16), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 1; }
    static unsigned long long h[kCands * 256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[kCands] = {"sqrt (rsq + Goldschmidt + residual)", "inv_sqrt2 A (2h seed, 2 Newton)", "inv_sqrt2 B (rcp seed, 2 Newton)", "inv_sqrt2 C (2h seed, 1 Newton)",
                                 "inv_sqrt2 D (2h seed, 3 Newton)", "rcp 1 Newton", "rcp 2 Newton"};
    for (int c = 0; c < kCands; ++c) {
        unsigned long long total = 0;
        int lo = -1, hi = -1, best_lo = 0, best_hi = -1;
        for (int e = 0; e < 256; ++e) {
            total += h[c * 256 + e];
            if (h[c * 256 + e] == 0) {
                if (lo < 0) lo = e;
                hi = e;
                if (hi - lo > best_hi - best_lo) best_lo = lo, best_hi = hi;
            } else {
                lo = -1;
            }
        }
        std::printf("%-40s mismatching inputs %llu | longest clean biased-exponent range [%d, %d]\n", names[c], total, best_lo, best_hi);
        std::printf("    per exponent (only non-zero):");
        int shown = 0;
        for (int e = 0; e < 256 && shown < 24; ++e)
            if (h[c * 256 + e]) std::printf(" %d:%llu", e, h[c * 256 + e]), ++shown;
        std::printf("\n");
    }
    return 0;
}
