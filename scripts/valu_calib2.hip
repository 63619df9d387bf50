// valu_calib2.hip — round 3 additions to the gfx950 VALU issue calibration (scripts/valu_calib.hip is round 2's table).
//
// Three questions:
//  1. Can the SQ counters measure VALU issue BUSY directly? Round 2 found that SQ_ACTIVE_INST_VALU counts one unit per 2- or 4-cycle
//     instruction, so it cannot tell the classes apart. gfx950 adds SQ_ACTIVE_INST_VALU2 ("quad-cycles two VALU instructions are
//     issued"): if it counts the pairs of full-rate instructions that share a quad-cycle, then 4 x (ACTIVE_INST_VALU - ACTIVE_INST_VALU2)
//     is the busy time — measured here on streams whose cycle cost is known from the shader clock.
//  2. Which of the SQ_INSTS_VALU_* type counters (ADD/MUL/FMA/TRANS_F32, CVT, INT32) count which opcodes of the traversal kernels?
//     They give the DYNAMIC instruction mix of the product kernels in the bench run itself.
//  3. Do gfx950's two-per-instruction conversions (v_cvt_pk_f32_fp8 / _bf8, v_cvt_scalef32_pk_f32_fp8 / _bf8 / _fp4) cost less than
//     2 cycles per value (the 24 v_cvt_f32_ubyteN of an inner step cost 4 each)? And: a byte masked IN PLACE is a denormal float
//     (N x 2^-149 for any integer N in bits 0..23) — does v_fma_f32 take a denormal multiplicand at full rate?
//
// Method as in valu_calib.hip: every wave issues ITERS x 128 instructions of one kind on 8 independent accumulators, 6 waves per
// SIMD (two 768-thread workgroups per CU, held apart by 64 KiB of LDS each); cycles = loop span on the chip-wide 100 MHz clock x the
// shader clock measured in the same loop. Build: hipcc -O2 --offload-arch=gfx950 scripts/valu_calib2.hip -o scripts/build/valu_calib2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

#define CHK(e)                                                               \
    do {                                                                     \
        hipError_t e_ = (e);                                                 \
        if (e_ != hipSuccess) {                                              \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));     \
            std::exit(2);                                                    \
        }                                                                    \
    } while (0)

enum Op {
    FMA, MUL, ADD_F32, MOV, AND, ADD_U32, LSHR, LSHL, CVT_UBYTE0, CVT_UBYTE2, MIN_F32, MAX3, CMP_E64, CNDMASK_E64, RCP, SQRT, CVT_F32_F16,
    ALT_FMA_CVT, ALT_FMA_FMA_CVT, ALT_MUL_AND,
    CVT_PK_FP8, CVT_PK_BF8, CVT_FP8, CVT_FP8_SDWA, SCALE_PK_FP8, SCALE_PK_BF8, SCALE_PK_FP4, SCALE_FP8, SCALE_PK_F16_FP8, CVT_OFF_I4,
    FMA_DENORM, AND_FMA_DENORM, AND_CVT_FMA,
    // pairing rules: which instruction forms can be the second instruction of a quad-cycle, and what breaks pairing
    FMA_SGPR, FMA_INLINE, FMA_LITERAL, MUL_LITERAL, ADDU_LITERAL, ALT_CMP_FMA, ALT_SEL_FMA, ALT_MIN_FMA, ALT_MAX3_FMA, ALT_CVT_AND, ALT_CVT_MOV, ALT_CVT_LSHR, ALT_CVT_ADDU,
    ALT_CVT_FMAC, ALT_CVT_FMA_SGPR, ALT_CVT_FMA_DEP, FMA_SALU, FMA_LDS, ALT_CVT_FMA_SALU, C3S1, C1S3, MIXED_WAVES, MIXED_WAVES_AND, N_OPS
};
static const char* kOpName[N_OPS] = {
    "v_fma_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_and_b32", "v_add_u32", "v_lshrrev_b32", "v_lshlrev_b32", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte2",
    "v_min_f32", "v_max3_f32", "v_cmp_lt_f32_e64 (sgpr pair)", "v_cndmask_b32_e64 (sgpr pair)", "v_rcp_f32", "v_sqrt_f32", "v_cvt_f32_f16",
    "alternating v_fma_f32 / v_cvt_f32_ubyte0 (1:1)", "alternating v_fma_f32, v_fma_f32, v_cvt_f32_ubyte0 (2:1)", "alternating v_mul_f32 / v_and_b32 (1:1, both full rate)",
    "v_cvt_pk_f32_fp8 (2 values)", "v_cvt_pk_f32_bf8 (2 values)", "v_cvt_f32_fp8 (byte 0)", "v_cvt_f32_fp8_sdwa (byte 2)", "v_cvt_scalef32_pk_f32_fp8 (2 values)",
    "v_cvt_scalef32_pk_f32_bf8 (2 values)", "v_cvt_scalef32_pk_f32_fp4 (2 values)", "v_cvt_scalef32_f32_fp8 (1 value)", "v_cvt_scalef32_pk_f16_fp8 (2 values, packed f16)",
    "v_cvt_off_f32_i4", "v_fma_f32 with a DENORMAL multiplicand (0x0000ab00)", "pair: v_and_b32 (byte in place) + v_fma_f32 on the denormal",
    "pair: v_cvt_f32_ubyte1 + v_fma_f32 (today's plane decode)",
    "v_fma_f32 with an SGPR multiplicand", "v_fma_f32 with an inline constant (2.0)", "v_fmac_f32 with a 32-bit literal (VOP3 takes none)", "v_mul_f32 with a 32-bit literal", "v_add_u32 with a 32-bit literal",
    "alternating v_cmp_lt_f32_e64 (sgpr pair) / v_fma_f32", "alternating v_cndmask_b32_e64 / v_fma_f32", "alternating v_min_f32 / v_fma_f32", "alternating v_max3_f32 / v_fma_f32",
    "alternating v_cvt_f32_ubyte0 / v_and_b32", "alternating v_cvt_f32_ubyte0 / v_mov_b32", "alternating v_cvt_f32_ubyte0 / v_lshrrev_b32", "alternating v_cvt_f32_ubyte0 / v_add_u32",
    "alternating v_cvt_f32_ubyte0 / v_fmac_f32 (VOP2)", "alternating v_cvt_f32_ubyte0 / v_fma_f32 with an SGPR multiplicand", "alternating v_cvt_f32_ubyte0 / v_fma_f32, the fma reading the cvt's result",
    "v_fma_f32 with an s_add_u32 after every instruction", "v_fma_f32 with a ds_read_b32 after every 4th instruction", "alternating v_cvt / v_fma with an s_add_u32 after every pair",
    "3 half-rate (v_cvt) : 1 full-rate (v_fma)", "1 half-rate (v_cvt) : 3 full-rate (v_fma)",
    "per SIMD 2 waves of pure v_cvt_f32_ubyte0 + 4 waves of pure v_fma_f32 (pairing ACROSS waves: 2.0; only inside a wave: 2.67)",
    "per SIMD 2 waves of pure v_cvt_f32_ubyte0 + 4 waves of pure v_and_b32"};
// VALU instructions per loop iteration (the asm bodies below)
static const double kInstrPerIter[N_OPS] = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
                                            128, 192, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 256, 256,
                                            128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
// values converted per instruction (conversion ops), for the cycles-per-value column
static const double kValuesPerInstr[N_OPS] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 2, 2, 1, 1, 2, 2, 2, 1, 2, 1, 0, 0, 0};  // (the pairing-rule streams: 0)

typedef float f2 __attribute__((ext_vector_type(2)));
#define ACC8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define PACC8 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
#define CLOB "vcc", "s20", "s21"
// one line per accumulator: X(n) expands to the instruction for accumulator %n; operands %8 = b (float), %9 = c (float), %10 = u (uint)
#define EIGHT(X) X("0") X("1") X("2") X("3") X("4") X("5") X("6") X("7")
#define BODY(X) asm volatile(REP16(EIGHT(X)) : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB)
#define PBODY(X) asm volatile(REP16(EIGHT(X)) : PACC8 : "v"(b), "v"(c), "v"(u) : CLOB)

#define I_FMA(n) "v_fma_f32 %" n ", %" n ", %8, %9\n\t"
#define I_MUL(n) "v_mul_f32 %" n ", %" n ", %8\n\t"
#define I_ADD(n) "v_add_f32 %" n ", %" n ", %9\n\t"
#define I_MOV(n) "v_mov_b32 %" n ", %8\n\t"
#define I_AND(n) "v_and_b32 %" n ", %" n ", %10\n\t"
#define I_ADDU(n) "v_add_u32 %" n ", %" n ", %10\n\t"
#define I_LSHR(n) "v_lshrrev_b32 %" n ", 1, %" n "\n\t"
#define I_LSHL(n) "v_lshlrev_b32 %" n ", 1, %" n "\n\t"
#define I_UB0(n) "v_cvt_f32_ubyte0 %" n ", %10\n\t"
#define I_UB2(n) "v_cvt_f32_ubyte2 %" n ", %10\n\t"
#define I_MIN(n) "v_min_f32 %" n ", %" n ", %8\n\t"
#define I_MAX3(n) "v_max3_f32 %" n ", %" n ", %8, %9\n\t"
#define I_CMP(n) "v_cmp_lt_f32_e64 s[20:21], %" n ", %8\n\t"
#define I_SEL(n) "v_cndmask_b32_e64 %" n ", %" n ", %8, s[20:21]\n\t"
#define I_RCP(n) "v_rcp_f32 %" n ", %" n "\n\t"
#define I_SQRT(n) "v_sqrt_f32 %" n ", %" n "\n\t"
#define I_F16(n) "v_cvt_f32_f16 %" n ", %10\n\t"
#define I_FMA_UB(n) "v_fma_f32 %" n ", %" n ", %8, %9\n\t"
#define I_PKFP8(n) "v_cvt_pk_f32_fp8 %" n ", %10\n\t"
#define I_PKBF8(n) "v_cvt_pk_f32_bf8 %" n ", %10\n\t"
#define I_FP8(n) "v_cvt_f32_fp8 %" n ", %10\n\t"
#define I_FP8S(n) "v_cvt_f32_fp8_sdwa %" n ", %10 src0_sel:BYTE_2\n\t"
#define I_SPKFP8(n) "v_cvt_scalef32_pk_f32_fp8 %" n ", %10, %8\n\t"
#define I_SPKBF8(n) "v_cvt_scalef32_pk_f32_bf8 %" n ", %10, %8\n\t"
#define I_SPKFP4(n) "v_cvt_scalef32_pk_f32_fp4 %" n ", %10, %8\n\t"
#define I_SFP8(n) "v_cvt_scalef32_f32_fp8 %" n ", %10, %8\n\t"
#define I_SPKF16(n) "v_cvt_scalef32_pk_f16_fp8 %" n ", %10, %8\n\t"
#define I_OFFI4(n) "v_cvt_off_f32_i4 %" n ", %10\n\t"
#define I_FMA_DEN(n) "v_fma_f32 %" n ", %10, %8, %" n "\n\t"
#define I_FMA_S(n) "v_fma_f32 %" n ", %" n ", s22, %9\n\t"
#define I_FMA_I(n) "v_fma_f32 %" n ", %" n ", 2.0, %9\n\t"
#define I_FMA_L(n) "v_fmac_f32 %" n ", 0x3f7fbe77, %9\n\t"
#define I_MUL_L(n) "v_mul_f32 %" n ", 0x3f7fbe77, %" n "\n\t"
#define I_ADDU_L(n) "v_add_u32 %" n ", 0x12345, %" n "\n\t"
#define I_FMAC(n) "v_fmac_f32 %" n ", %8, %9\n\t"
// alternating A (even accumulators) / B (odd accumulators)
#define ALT(A, B) asm volatile(REP16(A("0") B("1") A("2") B("3") A("4") B("5") A("6") B("7")) : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB)

template <int OP>
__global__ void __launch_bounds__(1024) k_calib2(uint32_t iters, float seed, unsigned long long* __restrict__ stamps, float* __restrict__ sink) {
    extern __shared__ uint32_t lds_pad[]; // occupancy control only
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b = 0.999f + seed * 1e-9f, c = 1e-7f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    uint32_t u = (OP == FMA_DENORM || OP == AND_FMA_DENORM) ? 0x0000ab00u : (0x38383838u | (__float_as_uint(seed) & 0x01010101u)); // fp8 0x38 = 1.0 (e4m3)
    float t0r = 0, t1r = 0, t2r = 0, t3r = 0; // scratch of the pair patterns
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("" ::: "memory");
    for (uint32_t i = 0; i < iters; ++i) {
        if (OP == FMA) BODY(I_FMA);
        else if (OP == MUL) BODY(I_MUL);
        else if (OP == ADD_F32) BODY(I_ADD);
        else if (OP == MOV) BODY(I_MOV);
        else if (OP == AND) BODY(I_AND);
        else if (OP == ADD_U32) BODY(I_ADDU);
        else if (OP == LSHR) BODY(I_LSHR);
        else if (OP == LSHL) BODY(I_LSHL);
        else if (OP == CVT_UBYTE0) BODY(I_UB0);
        else if (OP == CVT_UBYTE2) BODY(I_UB2);
        else if (OP == MIN_F32) BODY(I_MIN);
        else if (OP == MAX3) BODY(I_MAX3);
        else if (OP == CMP_E64) BODY(I_CMP);
        else if (OP == CNDMASK_E64) BODY(I_SEL);
        else if (OP == RCP) BODY(I_RCP);
        else if (OP == SQRT) BODY(I_SQRT);
        else if (OP == CVT_F32_F16) BODY(I_F16);
        else if (OP == ALT_FMA_CVT) { // 64 + 64
            asm volatile(REP16("v_fma_f32 %0, %0, %8, %9\n\t v_cvt_f32_ubyte0 %1, %10\n\t v_fma_f32 %2, %2, %8, %9\n\t v_cvt_f32_ubyte0 %3, %10\n\t"
                               "v_fma_f32 %4, %4, %8, %9\n\t v_cvt_f32_ubyte0 %5, %10\n\t v_fma_f32 %6, %6, %8, %9\n\t v_cvt_f32_ubyte0 %7, %10\n\t")
                         : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB);
        } else if (OP == ALT_FMA_FMA_CVT) { // 128 fma + 64 cvt
            asm volatile(REP16("v_fma_f32 %0, %0, %8, %9\n\t v_fma_f32 %1, %1, %8, %9\n\t v_cvt_f32_ubyte0 %2, %10\n\t v_fma_f32 %3, %3, %8, %9\n\t v_fma_f32 %4, %4, %8, %9\n\t"
                               "v_cvt_f32_ubyte0 %5, %10\n\t v_fma_f32 %6, %6, %8, %9\n\t v_fma_f32 %7, %7, %8, %9\n\t v_cvt_f32_ubyte0 %2, %10\n\t"
                               "v_fma_f32 %0, %0, %8, %9\n\t v_fma_f32 %1, %1, %8, %9\n\t v_cvt_f32_ubyte0 %5, %10\n\t")
                         : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB);
        } else if (OP == ALT_MUL_AND) {
            asm volatile(REP16("v_mul_f32 %0, %0, %8\n\t v_and_b32 %1, %1, %10\n\t v_mul_f32 %2, %2, %8\n\t v_and_b32 %3, %3, %10\n\t"
                               "v_mul_f32 %4, %4, %8\n\t v_and_b32 %5, %5, %10\n\t v_mul_f32 %6, %6, %8\n\t v_and_b32 %7, %7, %10\n\t")
                         : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB);
        }
        else if (OP == CVT_PK_FP8) PBODY(I_PKFP8);
        else if (OP == CVT_PK_BF8) PBODY(I_PKBF8);
        else if (OP == CVT_FP8) BODY(I_FP8);
        else if (OP == CVT_FP8_SDWA) BODY(I_FP8S);
        else if (OP == SCALE_PK_FP8) PBODY(I_SPKFP8);
        else if (OP == SCALE_PK_BF8) PBODY(I_SPKBF8);
        else if (OP == SCALE_PK_FP4) PBODY(I_SPKFP4);
        else if (OP == SCALE_FP8) BODY(I_SFP8);
        else if (OP == SCALE_PK_F16_FP8) BODY(I_SPKF16);
        else if (OP == CVT_OFF_I4) BODY(I_OFFI4);
        else if (OP == FMA_DENORM) BODY(I_FMA_DEN);
        else if (OP == AND_FMA_DENORM) { // the masked byte stays where it is: a denormal float; 4 x (and + fma) per line, 32 lines
            asm volatile(REP16(REP2("v_and_b32 %8, 0xff00, %14\n\t v_fma_f32 %0, %8, %12, %0\n\t v_and_b32 %9, 0xff00, %14\n\t v_fma_f32 %1, %9, %12, %1\n\t"
                                    "v_and_b32 %10, 0xff00, %14\n\t v_fma_f32 %2, %10, %12, %2\n\t v_and_b32 %11, 0xff00, %14\n\t v_fma_f32 %3, %11, %12, %3\n\t"))
                         : ACC8, "+v"(t0r), "+v"(t1r), "+v"(t2r), "+v"(t3r) : "v"(b), "v"(c), "v"(u) : CLOB);
        } else if (OP == AND_CVT_FMA) {
            asm volatile(REP16(REP2("v_cvt_f32_ubyte1 %8, %14\n\t v_fma_f32 %0, %8, %12, %0\n\t v_cvt_f32_ubyte1 %9, %14\n\t v_fma_f32 %1, %9, %12, %1\n\t"
                                    "v_cvt_f32_ubyte1 %10, %14\n\t v_fma_f32 %2, %10, %12, %2\n\t v_cvt_f32_ubyte1 %11, %14\n\t v_fma_f32 %3, %11, %12, %3\n\t"))
                         : ACC8, "+v"(t0r), "+v"(t1r), "+v"(t2r), "+v"(t3r) : "v"(b), "v"(c), "v"(u) : CLOB);
        }

        else if (OP == FMA_SGPR) { asm volatile("s_mov_b32 s22, 0x3f7fbe77" ::: "s22"); asm volatile(REP16(EIGHT(I_FMA_S)) : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB, "s22"); }
        else if (OP == FMA_INLINE) BODY(I_FMA_I);
        else if (OP == FMA_LITERAL) BODY(I_FMA_L);
        else if (OP == MUL_LITERAL) BODY(I_MUL_L);
        else if (OP == ADDU_LITERAL) BODY(I_ADDU_L);
        else if (OP == ALT_CMP_FMA) ALT(I_CMP, I_FMA);
        else if (OP == ALT_SEL_FMA) ALT(I_SEL, I_FMA);
        else if (OP == ALT_MIN_FMA) ALT(I_MIN, I_FMA);
        else if (OP == ALT_MAX3_FMA) ALT(I_MAX3, I_FMA);
        else if (OP == ALT_CVT_AND) ALT(I_UB0, I_AND);
        else if (OP == ALT_CVT_MOV) ALT(I_UB0, I_MOV);
        else if (OP == ALT_CVT_LSHR) ALT(I_UB0, I_LSHR);
        else if (OP == ALT_CVT_ADDU) ALT(I_UB0, I_ADDU);
        else if (OP == ALT_CVT_FMAC) ALT(I_UB0, I_FMAC);
        else if (OP == ALT_CVT_FMA_SGPR) { asm volatile("s_mov_b32 s22, 0x3f7fbe77" ::: "s22"); asm volatile(REP16(I_UB0("0") I_FMA_S("1") I_UB0("2") I_FMA_S("3") I_UB0("4") I_FMA_S("5") I_UB0("6") I_FMA_S("7")) : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB, "s22"); }
        else if (OP == ALT_CVT_FMA_DEP) asm volatile(REP16("v_cvt_f32_ubyte0 %0, %10\n\t v_fma_f32 %1, %0, %8, %9\n\t v_cvt_f32_ubyte0 %2, %10\n\t v_fma_f32 %3, %2, %8, %9\n\t"
                                                           "v_cvt_f32_ubyte0 %4, %10\n\t v_fma_f32 %5, %4, %8, %9\n\t v_cvt_f32_ubyte0 %6, %10\n\t v_fma_f32 %7, %6, %8, %9\n\t") : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB);
        else if (OP == FMA_SALU) asm volatile(REP16(I_FMA("0") "s_add_u32 s22, s22, 1\n\t" I_FMA("1") "s_add_u32 s22, s22, 1\n\t" I_FMA("2") "s_add_u32 s22, s22, 1\n\t" I_FMA("3") "s_add_u32 s22, s22, 1\n\t"
                                                    I_FMA("4") "s_add_u32 s22, s22, 1\n\t" I_FMA("5") "s_add_u32 s22, s22, 1\n\t" I_FMA("6") "s_add_u32 s22, s22, 1\n\t" I_FMA("7") "s_add_u32 s22, s22, 1\n\t")
                                              : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB, "s22", "scc");
        else if (OP == FMA_LDS) { asm volatile(REP16(I_FMA("0") I_FMA("1") I_FMA("2") I_FMA("3") "ds_read_b32 %11, %12\n\t" I_FMA("4") I_FMA("5") I_FMA("6") I_FMA("7") "ds_read_b32 %11, %12\n\t")
                                               "s_waitcnt lgkmcnt(0)\n\t" : ACC8 : "v"(b), "v"(c), "v"(u), "v"(t0r), "v"((threadIdx.x & 1023u) * 4u) : CLOB); }
        else if (OP == ALT_CVT_FMA_SALU) asm volatile(REP16(I_UB0("0") I_FMA("1") "s_add_u32 s22, s22, 1\n\t" I_UB0("2") I_FMA("3") "s_add_u32 s22, s22, 1\n\t"
                                                            I_UB0("4") I_FMA("5") "s_add_u32 s22, s22, 1\n\t" I_UB0("6") I_FMA("7") "s_add_u32 s22, s22, 1\n\t")
                                                      : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB, "s22", "scc");
        else if (OP == C3S1) asm volatile(REP16(I_UB0("0") I_UB0("1") I_UB0("2") I_FMA("3") I_UB0("4") I_UB0("5") I_UB0("6") I_FMA("7")) : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB);
        else if (OP == MIXED_WAVES) { // waves 4..7 of every 12-wave workgroup convert, the others multiply-add: every SIMD holds 2 + 4 of them
            if (((threadIdx.x >> 8) & 1u) != 0u) BODY(I_UB0); else BODY(I_FMA);
        } else if (OP == MIXED_WAVES_AND) {
            if (((threadIdx.x >> 8) & 1u) != 0u) BODY(I_UB0); else BODY(I_AND);
        }
        else if (OP == C1S3) asm volatile(REP16(I_UB0("0") I_FMA("1") I_FMA("2") I_FMA("3") I_UB0("4") I_FMA("5") I_FMA("6") I_FMA("7")) : ACC8 : "v"(b), "v"(c), "v"(u) : CLOB);
    }
    asm volatile("" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63u) == 0) stamps[4 * wave] = t0, stamps[4 * wave + 1] = t1, stamps[4 * wave + 2] = r0, stamps[4 * wave + 3] = r1;
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + t0r + t1r + t2r + t3r;
    if (s == 123.456f) sink[0] = s;
}

typedef void (*kern_t)(uint32_t, float, unsigned long long*, float*);
template <int... I>
static void fill(kern_t* k, std::integer_sequence<int, I...>) { ((k[I] = k_calib2<I>), ...); }

int main(int argc, char** argv) {
    const uint32_t iters = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 4000;
    CHK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    static kern_t kernels[N_OPS];
    fill(kernels, std::make_integer_sequence<int, N_OPS>{});
    unsigned long long* d_st;
    float* d_sink;
    const size_t max_waves = (size_t)n_cu * 32;
    CHK(hipMalloc(&d_st, max_waves * 4 * sizeof(unsigned long long)));
    CHK(hipMalloc(&d_sink, 64));
    std::vector<unsigned long long> st(max_waves * 4);
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %u, \"waves_per_simd\": 6, \"results\": [\n", prop.gcnArchName, n_cu, iters);
    const int n_blk = 2, w_blk = 3; // 6 waves per SIMD
    const size_t lds = 64 * 1024;
    for (int op = 0; op < N_OPS; ++op) {
        CHK(hipFuncSetAttribute((const void*)kernels[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const dim3 grid(n_cu * n_blk), block(256 * w_blk);
        const double ipi = kInstrPerIter[op];
        const uint32_t it = std::max<uint32_t>(8u, (uint32_t)((double)iters * 128.0 / ipi) / ((op == RCP || op == SQRT) ? 2u : 1u));
        hipLaunchKernelGGL(kernels[op], grid, block, lds, 0, it / 8 + 1, 1.0f, d_st, d_sink); // warm-up
        CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(kernels[op], grid, block, lds, 0, it, 1.0f, d_st, d_sink);
        CHK(hipDeviceSynchronize());
        CHK(hipGetLastError());
        const size_t n_waves = (size_t)grid.x * (block.x / 64);
        CHK(hipMemcpy(st.data(), d_st, n_waves * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> clk(n_waves);
        unsigned long long r_first = ~0ull, r_last = 0;
        for (size_t w = 0; w < n_waves; ++w) {
            clk[w] = (double)(st[4 * w + 1] - st[4 * w]) / std::max<double>(1.0, (double)(st[4 * w + 3] - st[4 * w + 2])) * 0.1; // GHz
            r_first = std::min(r_first, st[4 * w + 2]), r_last = std::max(r_last, st[4 * w + 3]);
        }
        std::sort(clk.begin(), clk.end());
        const double ghz = clk[n_waves / 2], span_s = (double)(r_last - r_first) * 1e-8;
        const double per_wave = (double)it * ipi;
        const double cpi = span_s * ghz * 1e9 / (per_wave * 6.0);
        std::printf("%s  {\"op_index\": %d, \"op\": \"%s\", \"cycles_per_instr_per_simd\": %.3f, \"cycles_per_value\": %s, \"shader_clock_ghz\": %.3f, \"valu_instructions\": %.0f, "
                    "\"simd_cycles\": %.0f}",
                    op ? ",\n" : "", op, kOpName[op], cpi, kValuesPerInstr[op] > 0 ? std::to_string(cpi / kValuesPerInstr[op]).c_str() : "null", ghz,
                    per_wave * (double)n_waves, span_s * ghz * 1e9);
    }
    std::printf("\n]}\n");
    return 0;
}
