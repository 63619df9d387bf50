// valu_calib.hip — calibration of the gfx950 VALU issue ceiling.
//
// Question: how many cycles does one SIMD need per wave64 VALU instruction, for the instruction kinds the traversal
// kernels are made of, with 1 / 2 / 4 / 6 / 8 waves resident per SIMD? The answer is the denominator of the
// "VALU issue busy" figure in profiles/rNN_valu.json, which round 1 ASSUMED to be 4 cycles.
//
// Method: every wave executes ITERS x 128 instructions of one kind on 8 independent accumulators (inline asm, so the
// compiler can neither fold nor re-associate them), stamps s_memtime before and after, and the host reports
//     cycles per instruction per SIMD = (last end - first start of the CU's waves) / (waves per SIMD x instructions per wave)
// from the in-kernel shader clock, plus the wall-clock figure from hipEvents. One workgroup per CU is forced by
// giving every workgroup 96 KiB of LDS (two never fit), its 256*W threads put W waves on each of the 4 SIMDs.
// The same binary is run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`
// (scripts/valu_calib.sh) so that the counters used on the render kernels are calibrated on a known instruction count.
//
// Build: hipcc -O2 --offload-arch=gfx950 scripts/valu_calib.hip -o gpurun_out/valu_calib   (standalone, no repo code)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "valu_mix_count.inc" // kMixCount_trav, kMixCount_mega, kMixHist_*

#define CHK(e)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (e);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                        \
            std::exit(2);                                                                       \
        }                                                                                       \
    } while (0)

enum Op { FMA, PK_FMA, CNDMASK, CVT_UBYTE, MAX3, MUL, MOV, RCP, FMA_DEP, FMA_HALF_EXEC, CMP, MIN_MAX,
          CNDMASK_VCC, PERM, BFE, AND, LSHL, LSHL_OR, ADD_U32, ADD_F32, CVT_F32_F16, CVT_F16_F32, CMP_VCC, MED3, XOR, MUL_LO, SQRT, MIX_TRAV, MIX_MEGA, N_OPS };
static const char* kOpName[N_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_cndmask_b32_e64 (sgpr pair)", "v_cvt_f32_ubyte1", "v_max3_f32", "v_mul_f32",
                                     "v_mov_b32", "v_rcp_f32", "v_fma_f32 (one dependent chain)", "v_fma_f32 (lanes 32-63 masked off)",
                                     "v_cmp_lt_f32_e64 (to sgpr pair)", "v_min_f32 + v_max_f32",
                                     "v_cndmask_b32_e32 (vcc)", "v_perm_b32", "v_bfe_u32", "v_and_b32", "v_lshlrev_b32", "v_lshl_or_b32", "v_add_u32",
                                     "v_add_f32", "v_cvt_f32_f16", "v_cvt_f16_f32", "v_cmp_lt_f32_e32 (to vcc)", "v_med3_f32", "v_xor_b32", "v_mul_lo_u32",
                                     "v_sqrt_f32", "static VALU mix of k_wf_extend (traversal loop)", "static VALU mix of k_megakernel"};

// 8 accumulators x 16 rounds = 128 instructions per loop iteration

template <int OP>
__global__ void __launch_bounds__(1024) k_calib(uint32_t iters, float seed, unsigned long long* __restrict__ stamps, float* __restrict__ sink) {
    extern __shared__ uint32_t lds_pad[]; // occupancy control only
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b = 0.999f + seed * 1e-9f, c = 1e-7f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 pb = {b, b}, pc = {c, c};
    uint32_t u = __float_as_uint(seed) | 0x00ff00u;
    if (OP == FMA_HALF_EXEC && (threadIdx.x & 32u)) return; // upper half of every wave leaves: exec = low 32 lanes
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("" ::: "memory");
    // ONE asm statement per loop iteration (128 instructions, 16 rounds over the 8 accumulators): the compiler puts nothing between them
#define ROUND8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BODY(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I) ROUND8(I)
#define ACC8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define PACC8 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
    for (uint32_t i = 0; i < iters; ++i) {
        if (OP == FMA || OP == FMA_HALF_EXEC) {
#define I(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == PK_FMA) {
#define I(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
            asm volatile(BODY(I) : PACC8 : "v"(pb), "v"(pc));
#undef I
        } else if (OP == CNDMASK) { // the mask is written once per 128 selects
#define I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n\t"
            asm volatile("v_cmp_lt_f32_e64 s[20:21], %8, %9\n\ts_nop 4\n\t" BODY(I) : ACC8 : "v"(b), "v"(c) : "s20", "s21");
#undef I
        } else if (OP == CNDMASK_VCC) {
#define I(k) "v_cndmask_b32_e32 %" #k ", %" #k ", %8, vcc\n\t"
            asm volatile("v_cmp_lt_f32_e32 vcc, %8, %9\n\ts_nop 4\n\t" BODY(I) : ACC8 : "v"(b), "v"(c) : "vcc");
#undef I
        } else if (OP == PERM) {
#define I(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(u));
#undef I
        } else if (OP == BFE) {
#define I(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == AND) {
#define I(k) "v_and_b32 %" #k ", %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == LSHL) {
#define I(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == LSHL_OR) {
#define I(k) "v_lshl_or_b32 %" #k ", %" #k ", 3, %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == ADD_U32) {
#define I(k) "v_add_u32 %" #k ", %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == ADD_F32) {
#define I(k) "v_add_f32 %" #k ", %" #k ", %9\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == CVT_F32_F16) {
#define I(k) "v_cvt_f32_f16 %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(u), "v"(c));
#undef I
        } else if (OP == CVT_F16_F32) {
#define I(k) "v_cvt_f16_f32 %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == CMP_VCC) {
#define I(k) "v_cmp_lt_f32_e32 vcc, %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c) : "vcc");
#undef I
        } else if (OP == MED3) {
#define I(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == XOR) {
#define I(k) "v_xor_b32 %" #k ", %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == MUL_LO) {
#define I(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(u), "v"(c));
#undef I
        } else if (OP == SQRT) {
#define I(k) "v_sqrt_f32 %" #k ", %" #k "\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == MIX_TRAV) { // kMixCount_trav instructions with the static VALU opcode mix of k_wf_extend (generated)
            asm volatile(
#include "valu_mix_body_trav.inc"
                : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23");
        } else if (OP == MIX_MEGA) {
            asm volatile(
#include "valu_mix_body_mega.inc"
                : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23");
        } else if (OP == CVT_UBYTE) {
#define I(k) "v_cvt_f32_ubyte1 %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(u), "v"(c));
#undef I
        } else if (OP == MAX3) {
#define I(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == MUL) {
#define I(k) "v_mul_f32 %" #k ", %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == MOV) {
#define I(k) "v_mov_b32 %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == RCP) {
#define I(k) "v_rcp_f32 %" #k ", %" #k "\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == FMA_DEP) {
#define I(k) "v_fma_f32 %0, %0, %8, %9\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        } else if (OP == CMP) {
#define I(k) "v_cmp_lt_f32 s[20:21], %" #k ", %8\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c) : "s20", "s21");
#undef I
        } else if (OP == MIN_MAX) { // 256 instructions per iteration: the host halves the figure
#define I(k) "v_min_f32 %" #k ", %" #k ", %8\n\tv_max_f32 %" #k ", %" #k ", %9\n\t"
            asm volatile(BODY(I) : ACC8 : "v"(b), "v"(c));
#undef I
        }
    }
    asm volatile("" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63u) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[4 * w] = t0, stamps[4 * w + 1] = t1, stamps[4 * w + 2] = r0, stamps[4 * w + 3] = r1;
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (s == 12345.678f) sink[0] = s; // keeps every accumulator alive
}

typedef void (*kern_t)(uint32_t, float, unsigned long long*, float*);
template <int OP>
kern_t kern() { return k_calib<OP>; }

int main(int argc, char** argv) {
    const uint32_t iters = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 4000;
    int dev = 0;
    CHK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, dev));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long* d_st;
    float* d_sink;
    const size_t max_waves = (size_t)n_cu * 32;
    CHK(hipMalloc(&d_st, max_waves * 4 * sizeof(unsigned long long)));
    CHK(hipMalloc(&d_sink, 64));
    std::vector<unsigned long long> st(max_waves * 4);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    kern_t ks[N_OPS] = {kern<FMA>(),     kern<PK_FMA>(), kern<CNDMASK>(),       kern<CVT_UBYTE>(), kern<MAX3>(),   kern<MUL>(),
                        kern<MOV>(),     kern<RCP>(),    kern<FMA_DEP>(),       kern<FMA_HALF_EXEC>(), kern<CMP>(), kern<MIN_MAX>(),
                        kern<CNDMASK_VCC>(), kern<PERM>(), kern<BFE>(), kern<AND>(), kern<LSHL>(), kern<LSHL_OR>(), kern<ADD_U32>(), kern<ADD_F32>(),
                        kern<CVT_F32_F16>(), kern<CVT_F16_F32>(), kern<CMP_VCC>(), kern<MED3>(), kern<XOR>(), kern<MUL_LO>(), kern<SQRT>(), kern<MIX_TRAV>(), kern<MIX_MEGA>()};
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iters\": %u, \"instr_per_wave\": %llu, \"results\": [\n", prop.gcnArchName, n_cu,
                prop.clockRate / 1000, iters, (unsigned long long)iters * 128ull);
    bool first = true;
    const int wps_list[] = {1, 2, 3, 4, 6, 8};
    for (int op = 0; op < N_OPS; ++op) {
        for (int wps : wps_list) {
            // W waves per SIMD: blocks of 256*w_blk threads, n_blk per CU; LDS keeps n_blk blocks per CU and no more
            const int n_blk = wps > 4 ? 2 : 1, w_blk = wps / n_blk; // 6 -> 2 x 3, 8 -> 2 x 4
            const size_t lds = n_blk == 1 ? 96 * 1024 : 64 * 1024;
            CHK(hipFuncSetAttribute((const void*)ks[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const dim3 grid(n_cu * n_blk), block(256 * w_blk);
            const uint32_t it = (op == RCP || op == SQRT) ? iters / 2 : op == MIX_TRAV ? iters / 4 : op == MIX_MEGA ? iters / 16 : iters;
            hipLaunchKernelGGL(ks[op], grid, block, lds, 0, it / 8 + 1, 1.0f, d_st, d_sink); // warm-up
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(ks[op], grid, block, lds, 0, it, 1.0f, d_st, d_sink);
            CHK(hipEventRecord(e1, 0));
            CHK(hipDeviceSynchronize());
            CHK(hipGetLastError());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const size_t n_waves = (size_t)grid.x * (block.x / 64);
            CHK(hipMemcpy(st.data(), d_st, n_waves * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            // shader clock the chip held during the loop: s_memtime ticks per 100 MHz s_memrealtime tick, median over the waves;
            // loop span = first wave in -> last wave out on the chip-wide 100 MHz clock (no launch overhead in it)
            std::vector<double> clk(n_waves), dur(n_waves);
            unsigned long long r_first = ~0ull, r_last = 0;
            for (size_t w = 0; w < n_waves; ++w) {
                dur[w] = (double)(st[4 * w + 1] - st[4 * w]);
                clk[w] = dur[w] / std::max<double>(1.0, (double)(st[4 * w + 3] - st[4 * w + 2])) * 0.1; // GHz
                r_first = std::min(r_first, st[4 * w + 2]), r_last = std::max(r_last, st[4 * w + 3]);
            }
            std::sort(clk.begin(), clk.end());
            std::sort(dur.begin(), dur.end());
            const double ghz = clk[n_waves / 2], span_s = (double)(r_last - r_first) * 1e-8;
            const double per_wave = (double)it * (op == MIN_MAX ? 256.0 : op == MIX_TRAV ? (double)kMixCount_trav : op == MIX_MEGA ? (double)kMixCount_mega : 128.0);
            const double cpi_simd = span_s * ghz * 1e9 / (per_wave * wps); // every SIMD issues per_wave * wps instructions during the span
            std::printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_simd\": %.3f, \"shader_clock_ghz\": %.3f, \"loop_span_ms\": %.4f, "
                        "\"one_wave_cycles_per_instr\": %.3f, \"kernel_ms_hipevents\": %.4f, \"valu_instructions\": %.0f}",
                        first ? "" : ",\n", kOpName[op], wps, cpi_simd, ghz, span_s * 1e3, dur[n_waves / 2] / per_wave, ms, per_wave * (double)n_waves);
            first = false;
        }
    }
    std::printf("\n]}\n");
    return 0;
}
