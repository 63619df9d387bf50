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

// REPn(x): the string literal(s) x, n times (the asm bodies below are one period of their instruction pattern, repeated)
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)
#define REP64(x) REP32(x) REP32(x)
#define REP128(x) REP64(x) REP64(x)

#define CHK(e)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (e);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                        \
            std::exit(2);                                                                       \
        }                                                                                       \
    } while (0)

// One entry per instruction kind: 16 rounds over 8 accumulators = 128 instructions per loop iteration unless kInstrPerIter says otherwise
enum Op { FMA, MUL, ADD_F32, SUB_F32, FMAC, MOV, AND, OR, XOR, ADD_U32, SUB_U32, MUL_U24, MAD_U24, FMA_DEP, FMA_HALF_EXEC, FMA_MIX, FMA_MIX_HI, PK_FMA, PK_MUL, PK_ADD, PK_FMA_F16, PK_MAX_F16, LSHL, LSHR, ASHR, LSHL_OR, LSHL_ADD, AND_OR, ADD3, BFE, BFI, PERM, ALIGNBIT, MUL_LO, MIN_F32, MAX_F32, MIN_U32, MAX3, MED3, CVT_UBYTE, CVT_F32_U32, CVT_U32_F32, CVT_F32_F16, CVT_F16_F32, FLOOR, LDEXP, RCP, SQRT, RSQ, DIV_SCALE, DIV_FMAS, DIV_FIXUP, CMP_E64, CMP_E32, CMP_U32_E64, CMPX, CNDMASK_E64, CNDMASK_E64_VCC, CNDMASK_E32, PAIR_E32, PAIR_E64, CMP_4SEL_E64, CMP_4SEL_VCC, EXEC_MOV, SWAP, READFIRSTLANE, MBCNT, MOV_DPP, MOV_SDWA, OR_SDWA, ADD_F32_SDWA, CVT_F32_F16_SDWA, CVT_UBYTE3, MUL_F32_ABS, MAX_F32_E64, LSHL_ADD_U64, MIX_TRAV, MIX_MEGA, N_OPS };
static const char* kOpName[N_OPS] = {
    "v_fma_f32",
    "v_mul_f32",
    "v_add_f32",
    "v_sub_f32",
    "v_fmac_f32",
    "v_mov_b32",
    "v_and_b32",
    "v_or_b32",
    "v_xor_b32",
    "v_add_u32",
    "v_sub_u32",
    "v_mul_u32_u24",
    "v_mad_u32_u24",
    "v_fma_f32 (one dependent chain)",
    "v_fma_f32 (lanes 32-63 masked off)",
    "v_fma_mix_f32 (src0 f16 lo)",
    "v_fma_mix_f32 (src0 f16 hi)",
    "v_pk_fma_f32",
    "v_pk_mul_f32",
    "v_pk_add_f32",
    "v_pk_fma_f16",
    "v_pk_max_f16",
    "v_lshlrev_b32",
    "v_lshrrev_b32",
    "v_ashrrev_i32",
    "v_lshl_or_b32",
    "v_lshl_add_u32",
    "v_and_or_b32",
    "v_add3_u32",
    "v_bfe_u32",
    "v_bfi_b32",
    "v_perm_b32",
    "v_alignbit_b32",
    "v_mul_lo_u32",
    "v_min_f32",
    "v_max_f32",
    "v_min_u32",
    "v_max3_f32",
    "v_med3_f32",
    "v_cvt_f32_ubyte1",
    "v_cvt_f32_u32",
    "v_cvt_u32_f32",
    "v_cvt_f32_f16",
    "v_cvt_f16_f32",
    "v_floor_f32",
    "v_ldexp_f32",
    "v_rcp_f32",
    "v_sqrt_f32",
    "v_rsq_f32",
    "v_div_scale_f32",
    "v_div_fmas_f32",
    "v_div_fixup_f32",
    "v_cmp_lt_f32_e64 (to sgpr pair)",
    "v_cmp_lt_f32_e32 (to vcc)",
    "v_cmp_lt_u32_e64 (to sgpr pair)",
    "v_cmpx_le_f32_e64 (exec; a always-true compare)",
    "v_cndmask_b32_e64 (sgpr pair, written once per 128)",
    "v_cndmask_b32_e64 (vcc as explicit operand, written once per 128)",
    "v_cndmask_b32_e32 (vcc, written once per 128)",
    "pair: v_cmp_lt_f32_e32 vcc + v_cndmask_b32_e32 vcc",
    "pair: v_cmp_lt_f32_e64 sgpr + v_cndmask_b32_e64 sgpr",
    "group: v_cmp_e64 + 4 v_cndmask_b32_e64 on that mask (one compare-exchange of the sort)",
    "group: v_cmp_e32 + 4 v_cndmask_b32_e32 on vcc",
    "group: s_mov exec, mask + 4 v_mov_b32 + s_mov exec, -1 (select by exec mask)",
    "v_swap_b32",
    "v_readfirstlane_b32",
    "v_mbcnt_lo_u32_b32",
    "v_mov_b32_dpp (row_shr:1)",
    "v_mov_b32_sdwa (byte 2 of src -> byte 1 of dst, rest preserved)",
    "v_or_b32_sdwa (src0 byte select)",
    "v_add_f32_sdwa (src0 word select)",
    "v_cvt_f32_f16_sdwa (hi half)",
    "v_cvt_f32_ubyte3",
    "v_mul_f32_e64 with |abs| modifier",
    "v_max_f32_e64 (neg modifier)",
    "v_lshl_add_u64",
    "static VALU mix of k_wf_extend (traversal loop)",
    "static VALU mix of k_megakernel"};

#define ACC8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define PACC8 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
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
    // ONE asm statement per loop iteration: the compiler puts nothing between the measured instructions
    for (uint32_t i = 0; i < iters; ++i) {
        if (OP == FMA) {
            asm volatile(REP16(
            "v_fma_f32 %0, %0, %8, %9\n\t"
            "v_fma_f32 %1, %1, %8, %9\n\t"
            "v_fma_f32 %2, %2, %8, %9\n\t"
            "v_fma_f32 %3, %3, %8, %9\n\t"
            "v_fma_f32 %4, %4, %8, %9\n\t"
            "v_fma_f32 %5, %5, %8, %9\n\t"
            "v_fma_f32 %6, %6, %8, %9\n\t"
            "v_fma_f32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MUL) {
            asm volatile(REP16(
            "v_mul_f32 %0, %0, %8\n\t"
            "v_mul_f32 %1, %1, %8\n\t"
            "v_mul_f32 %2, %2, %8\n\t"
            "v_mul_f32 %3, %3, %8\n\t"
            "v_mul_f32 %4, %4, %8\n\t"
            "v_mul_f32 %5, %5, %8\n\t"
            "v_mul_f32 %6, %6, %8\n\t"
            "v_mul_f32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == ADD_F32) {
            asm volatile(REP16(
            "v_add_f32 %0, %0, %9\n\t"
            "v_add_f32 %1, %1, %9\n\t"
            "v_add_f32 %2, %2, %9\n\t"
            "v_add_f32 %3, %3, %9\n\t"
            "v_add_f32 %4, %4, %9\n\t"
            "v_add_f32 %5, %5, %9\n\t"
            "v_add_f32 %6, %6, %9\n\t"
            "v_add_f32 %7, %7, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == SUB_F32) {
            asm volatile(REP16(
            "v_sub_f32 %0, %0, %9\n\t"
            "v_sub_f32 %1, %1, %9\n\t"
            "v_sub_f32 %2, %2, %9\n\t"
            "v_sub_f32 %3, %3, %9\n\t"
            "v_sub_f32 %4, %4, %9\n\t"
            "v_sub_f32 %5, %5, %9\n\t"
            "v_sub_f32 %6, %6, %9\n\t"
            "v_sub_f32 %7, %7, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == FMAC) {
            asm volatile(REP16(
            "v_fmac_f32 %0, %8, %9\n\t"
            "v_fmac_f32 %1, %8, %9\n\t"
            "v_fmac_f32 %2, %8, %9\n\t"
            "v_fmac_f32 %3, %8, %9\n\t"
            "v_fmac_f32 %4, %8, %9\n\t"
            "v_fmac_f32 %5, %8, %9\n\t"
            "v_fmac_f32 %6, %8, %9\n\t"
            "v_fmac_f32 %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MOV) {
            asm volatile(REP16(
            "v_mov_b32 %0, %8\n\t"
            "v_mov_b32 %1, %8\n\t"
            "v_mov_b32 %2, %8\n\t"
            "v_mov_b32 %3, %8\n\t"
            "v_mov_b32 %4, %8\n\t"
            "v_mov_b32 %5, %8\n\t"
            "v_mov_b32 %6, %8\n\t"
            "v_mov_b32 %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == AND) {
            asm volatile(REP16(
            "v_and_b32 %0, %0, %8\n\t"
            "v_and_b32 %1, %1, %8\n\t"
            "v_and_b32 %2, %2, %8\n\t"
            "v_and_b32 %3, %3, %8\n\t"
            "v_and_b32 %4, %4, %8\n\t"
            "v_and_b32 %5, %5, %8\n\t"
            "v_and_b32 %6, %6, %8\n\t"
            "v_and_b32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == OR) {
            asm volatile(REP16(
            "v_or_b32 %0, %0, %8\n\t"
            "v_or_b32 %1, %1, %8\n\t"
            "v_or_b32 %2, %2, %8\n\t"
            "v_or_b32 %3, %3, %8\n\t"
            "v_or_b32 %4, %4, %8\n\t"
            "v_or_b32 %5, %5, %8\n\t"
            "v_or_b32 %6, %6, %8\n\t"
            "v_or_b32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == XOR) {
            asm volatile(REP16(
            "v_xor_b32 %0, %0, %8\n\t"
            "v_xor_b32 %1, %1, %8\n\t"
            "v_xor_b32 %2, %2, %8\n\t"
            "v_xor_b32 %3, %3, %8\n\t"
            "v_xor_b32 %4, %4, %8\n\t"
            "v_xor_b32 %5, %5, %8\n\t"
            "v_xor_b32 %6, %6, %8\n\t"
            "v_xor_b32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == ADD_U32) {
            asm volatile(REP16(
            "v_add_u32 %0, %0, %8\n\t"
            "v_add_u32 %1, %1, %8\n\t"
            "v_add_u32 %2, %2, %8\n\t"
            "v_add_u32 %3, %3, %8\n\t"
            "v_add_u32 %4, %4, %8\n\t"
            "v_add_u32 %5, %5, %8\n\t"
            "v_add_u32 %6, %6, %8\n\t"
            "v_add_u32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == SUB_U32) {
            asm volatile(REP16(
            "v_sub_u32 %0, %0, %8\n\t"
            "v_sub_u32 %1, %1, %8\n\t"
            "v_sub_u32 %2, %2, %8\n\t"
            "v_sub_u32 %3, %3, %8\n\t"
            "v_sub_u32 %4, %4, %8\n\t"
            "v_sub_u32 %5, %5, %8\n\t"
            "v_sub_u32 %6, %6, %8\n\t"
            "v_sub_u32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MUL_U24) {
            asm volatile(REP16(
            "v_mul_u32_u24 %0, %0, %8\n\t"
            "v_mul_u32_u24 %1, %1, %8\n\t"
            "v_mul_u32_u24 %2, %2, %8\n\t"
            "v_mul_u32_u24 %3, %3, %8\n\t"
            "v_mul_u32_u24 %4, %4, %8\n\t"
            "v_mul_u32_u24 %5, %5, %8\n\t"
            "v_mul_u32_u24 %6, %6, %8\n\t"
            "v_mul_u32_u24 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MAD_U24) {
            asm volatile(REP16(
            "v_mad_u32_u24 %0, %0, %8, %9\n\t"
            "v_mad_u32_u24 %1, %1, %8, %9\n\t"
            "v_mad_u32_u24 %2, %2, %8, %9\n\t"
            "v_mad_u32_u24 %3, %3, %8, %9\n\t"
            "v_mad_u32_u24 %4, %4, %8, %9\n\t"
            "v_mad_u32_u24 %5, %5, %8, %9\n\t"
            "v_mad_u32_u24 %6, %6, %8, %9\n\t"
            "v_mad_u32_u24 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == FMA_DEP) {
            asm volatile(REP128(
            "v_fma_f32 %0, %0, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == FMA_HALF_EXEC) {
            asm volatile(REP16(
            "v_fma_f32 %0, %0, %8, %9\n\t"
            "v_fma_f32 %1, %1, %8, %9\n\t"
            "v_fma_f32 %2, %2, %8, %9\n\t"
            "v_fma_f32 %3, %3, %8, %9\n\t"
            "v_fma_f32 %4, %4, %8, %9\n\t"
            "v_fma_f32 %5, %5, %8, %9\n\t"
            "v_fma_f32 %6, %6, %8, %9\n\t"
            "v_fma_f32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == FMA_MIX) {
            asm volatile(REP16(
            "v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[1,0,0]\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == FMA_MIX_HI) {
            asm volatile(REP16(
            "v_fma_mix_f32 %0, %0, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %1, %1, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %2, %2, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %3, %3, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %4, %4, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %5, %5, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %6, %6, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %7, %7, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == PK_FMA) {
            asm volatile(REP16(
            "v_pk_fma_f32 %0, %0, %8, %9\n\t"
            "v_pk_fma_f32 %1, %1, %8, %9\n\t"
            "v_pk_fma_f32 %2, %2, %8, %9\n\t"
            "v_pk_fma_f32 %3, %3, %8, %9\n\t"
            "v_pk_fma_f32 %4, %4, %8, %9\n\t"
            "v_pk_fma_f32 %5, %5, %8, %9\n\t"
            "v_pk_fma_f32 %6, %6, %8, %9\n\t"
            "v_pk_fma_f32 %7, %7, %8, %9\n\t"
            )
              : PACC8 : "v"(pb), "v"(pc));
        }
        else if (OP == PK_MUL) {
            asm volatile(REP16(
            "v_pk_mul_f32 %0, %0, %8\n\t"
            "v_pk_mul_f32 %1, %1, %8\n\t"
            "v_pk_mul_f32 %2, %2, %8\n\t"
            "v_pk_mul_f32 %3, %3, %8\n\t"
            "v_pk_mul_f32 %4, %4, %8\n\t"
            "v_pk_mul_f32 %5, %5, %8\n\t"
            "v_pk_mul_f32 %6, %6, %8\n\t"
            "v_pk_mul_f32 %7, %7, %8\n\t"
            )
              : PACC8 : "v"(pb), "v"(pc));
        }
        else if (OP == PK_ADD) {
            asm volatile(REP16(
            "v_pk_add_f32 %0, %0, %8\n\t"
            "v_pk_add_f32 %1, %1, %8\n\t"
            "v_pk_add_f32 %2, %2, %8\n\t"
            "v_pk_add_f32 %3, %3, %8\n\t"
            "v_pk_add_f32 %4, %4, %8\n\t"
            "v_pk_add_f32 %5, %5, %8\n\t"
            "v_pk_add_f32 %6, %6, %8\n\t"
            "v_pk_add_f32 %7, %7, %8\n\t"
            )
              : PACC8 : "v"(pb), "v"(pc));
        }
        else if (OP == PK_FMA_F16) {
            asm volatile(REP16(
            "v_pk_fma_f16 %0, %0, %8, %9\n\t"
            "v_pk_fma_f16 %1, %1, %8, %9\n\t"
            "v_pk_fma_f16 %2, %2, %8, %9\n\t"
            "v_pk_fma_f16 %3, %3, %8, %9\n\t"
            "v_pk_fma_f16 %4, %4, %8, %9\n\t"
            "v_pk_fma_f16 %5, %5, %8, %9\n\t"
            "v_pk_fma_f16 %6, %6, %8, %9\n\t"
            "v_pk_fma_f16 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == PK_MAX_F16) {
            asm volatile(REP16(
            "v_pk_max_f16 %0, %0, %8\n\t"
            "v_pk_max_f16 %1, %1, %8\n\t"
            "v_pk_max_f16 %2, %2, %8\n\t"
            "v_pk_max_f16 %3, %3, %8\n\t"
            "v_pk_max_f16 %4, %4, %8\n\t"
            "v_pk_max_f16 %5, %5, %8\n\t"
            "v_pk_max_f16 %6, %6, %8\n\t"
            "v_pk_max_f16 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == LSHL) {
            asm volatile(REP16(
            "v_lshlrev_b32 %0, 1, %0\n\t"
            "v_lshlrev_b32 %1, 1, %1\n\t"
            "v_lshlrev_b32 %2, 1, %2\n\t"
            "v_lshlrev_b32 %3, 1, %3\n\t"
            "v_lshlrev_b32 %4, 1, %4\n\t"
            "v_lshlrev_b32 %5, 1, %5\n\t"
            "v_lshlrev_b32 %6, 1, %6\n\t"
            "v_lshlrev_b32 %7, 1, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == LSHR) {
            asm volatile(REP16(
            "v_lshrrev_b32 %0, 1, %0\n\t"
            "v_lshrrev_b32 %1, 1, %1\n\t"
            "v_lshrrev_b32 %2, 1, %2\n\t"
            "v_lshrrev_b32 %3, 1, %3\n\t"
            "v_lshrrev_b32 %4, 1, %4\n\t"
            "v_lshrrev_b32 %5, 1, %5\n\t"
            "v_lshrrev_b32 %6, 1, %6\n\t"
            "v_lshrrev_b32 %7, 1, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == ASHR) {
            asm volatile(REP16(
            "v_ashrrev_i32 %0, 1, %0\n\t"
            "v_ashrrev_i32 %1, 1, %1\n\t"
            "v_ashrrev_i32 %2, 1, %2\n\t"
            "v_ashrrev_i32 %3, 1, %3\n\t"
            "v_ashrrev_i32 %4, 1, %4\n\t"
            "v_ashrrev_i32 %5, 1, %5\n\t"
            "v_ashrrev_i32 %6, 1, %6\n\t"
            "v_ashrrev_i32 %7, 1, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == LSHL_OR) {
            asm volatile(REP16(
            "v_lshl_or_b32 %0, %0, 3, %8\n\t"
            "v_lshl_or_b32 %1, %1, 3, %8\n\t"
            "v_lshl_or_b32 %2, %2, 3, %8\n\t"
            "v_lshl_or_b32 %3, %3, 3, %8\n\t"
            "v_lshl_or_b32 %4, %4, 3, %8\n\t"
            "v_lshl_or_b32 %5, %5, 3, %8\n\t"
            "v_lshl_or_b32 %6, %6, 3, %8\n\t"
            "v_lshl_or_b32 %7, %7, 3, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == LSHL_ADD) {
            asm volatile(REP16(
            "v_lshl_add_u32 %0, %0, 3, %8\n\t"
            "v_lshl_add_u32 %1, %1, 3, %8\n\t"
            "v_lshl_add_u32 %2, %2, 3, %8\n\t"
            "v_lshl_add_u32 %3, %3, 3, %8\n\t"
            "v_lshl_add_u32 %4, %4, 3, %8\n\t"
            "v_lshl_add_u32 %5, %5, 3, %8\n\t"
            "v_lshl_add_u32 %6, %6, 3, %8\n\t"
            "v_lshl_add_u32 %7, %7, 3, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == AND_OR) {
            asm volatile(REP16(
            "v_and_or_b32 %0, %0, %8, %9\n\t"
            "v_and_or_b32 %1, %1, %8, %9\n\t"
            "v_and_or_b32 %2, %2, %8, %9\n\t"
            "v_and_or_b32 %3, %3, %8, %9\n\t"
            "v_and_or_b32 %4, %4, %8, %9\n\t"
            "v_and_or_b32 %5, %5, %8, %9\n\t"
            "v_and_or_b32 %6, %6, %8, %9\n\t"
            "v_and_or_b32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == ADD3) {
            asm volatile(REP16(
            "v_add3_u32 %0, %0, %8, %9\n\t"
            "v_add3_u32 %1, %1, %8, %9\n\t"
            "v_add3_u32 %2, %2, %8, %9\n\t"
            "v_add3_u32 %3, %3, %8, %9\n\t"
            "v_add3_u32 %4, %4, %8, %9\n\t"
            "v_add3_u32 %5, %5, %8, %9\n\t"
            "v_add3_u32 %6, %6, %8, %9\n\t"
            "v_add3_u32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == BFE) {
            asm volatile(REP16(
            "v_bfe_u32 %0, %0, 8, 8\n\t"
            "v_bfe_u32 %1, %1, 8, 8\n\t"
            "v_bfe_u32 %2, %2, 8, 8\n\t"
            "v_bfe_u32 %3, %3, 8, 8\n\t"
            "v_bfe_u32 %4, %4, 8, 8\n\t"
            "v_bfe_u32 %5, %5, 8, 8\n\t"
            "v_bfe_u32 %6, %6, 8, 8\n\t"
            "v_bfe_u32 %7, %7, 8, 8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == BFI) {
            asm volatile(REP16(
            "v_bfi_b32 %0, %0, %8, %9\n\t"
            "v_bfi_b32 %1, %1, %8, %9\n\t"
            "v_bfi_b32 %2, %2, %8, %9\n\t"
            "v_bfi_b32 %3, %3, %8, %9\n\t"
            "v_bfi_b32 %4, %4, %8, %9\n\t"
            "v_bfi_b32 %5, %5, %8, %9\n\t"
            "v_bfi_b32 %6, %6, %8, %9\n\t"
            "v_bfi_b32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == PERM) {
            asm volatile(REP16(
            "v_perm_b32 %0, %0, %8, %9\n\t"
            "v_perm_b32 %1, %1, %8, %9\n\t"
            "v_perm_b32 %2, %2, %8, %9\n\t"
            "v_perm_b32 %3, %3, %8, %9\n\t"
            "v_perm_b32 %4, %4, %8, %9\n\t"
            "v_perm_b32 %5, %5, %8, %9\n\t"
            "v_perm_b32 %6, %6, %8, %9\n\t"
            "v_perm_b32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(u), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == ALIGNBIT) {
            asm volatile(REP16(
            "v_alignbit_b32 %0, %0, %8, 8\n\t"
            "v_alignbit_b32 %1, %1, %8, 8\n\t"
            "v_alignbit_b32 %2, %2, %8, 8\n\t"
            "v_alignbit_b32 %3, %3, %8, 8\n\t"
            "v_alignbit_b32 %4, %4, %8, 8\n\t"
            "v_alignbit_b32 %5, %5, %8, 8\n\t"
            "v_alignbit_b32 %6, %6, %8, 8\n\t"
            "v_alignbit_b32 %7, %7, %8, 8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MUL_LO) {
            asm volatile(REP16(
            "v_mul_lo_u32 %0, %0, %8\n\t"
            "v_mul_lo_u32 %1, %1, %8\n\t"
            "v_mul_lo_u32 %2, %2, %8\n\t"
            "v_mul_lo_u32 %3, %3, %8\n\t"
            "v_mul_lo_u32 %4, %4, %8\n\t"
            "v_mul_lo_u32 %5, %5, %8\n\t"
            "v_mul_lo_u32 %6, %6, %8\n\t"
            "v_mul_lo_u32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MIN_F32) {
            asm volatile(REP16(
            "v_min_f32 %0, %0, %8\n\t"
            "v_min_f32 %1, %1, %8\n\t"
            "v_min_f32 %2, %2, %8\n\t"
            "v_min_f32 %3, %3, %8\n\t"
            "v_min_f32 %4, %4, %8\n\t"
            "v_min_f32 %5, %5, %8\n\t"
            "v_min_f32 %6, %6, %8\n\t"
            "v_min_f32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MAX_F32) {
            asm volatile(REP16(
            "v_max_f32 %0, %0, %9\n\t"
            "v_max_f32 %1, %1, %9\n\t"
            "v_max_f32 %2, %2, %9\n\t"
            "v_max_f32 %3, %3, %9\n\t"
            "v_max_f32 %4, %4, %9\n\t"
            "v_max_f32 %5, %5, %9\n\t"
            "v_max_f32 %6, %6, %9\n\t"
            "v_max_f32 %7, %7, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MIN_U32) {
            asm volatile(REP16(
            "v_min_u32 %0, %0, %8\n\t"
            "v_min_u32 %1, %1, %8\n\t"
            "v_min_u32 %2, %2, %8\n\t"
            "v_min_u32 %3, %3, %8\n\t"
            "v_min_u32 %4, %4, %8\n\t"
            "v_min_u32 %5, %5, %8\n\t"
            "v_min_u32 %6, %6, %8\n\t"
            "v_min_u32 %7, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MAX3) {
            asm volatile(REP16(
            "v_max3_f32 %0, %0, %8, %9\n\t"
            "v_max3_f32 %1, %1, %8, %9\n\t"
            "v_max3_f32 %2, %2, %8, %9\n\t"
            "v_max3_f32 %3, %3, %8, %9\n\t"
            "v_max3_f32 %4, %4, %8, %9\n\t"
            "v_max3_f32 %5, %5, %8, %9\n\t"
            "v_max3_f32 %6, %6, %8, %9\n\t"
            "v_max3_f32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MED3) {
            asm volatile(REP16(
            "v_med3_f32 %0, %0, %8, %9\n\t"
            "v_med3_f32 %1, %1, %8, %9\n\t"
            "v_med3_f32 %2, %2, %8, %9\n\t"
            "v_med3_f32 %3, %3, %8, %9\n\t"
            "v_med3_f32 %4, %4, %8, %9\n\t"
            "v_med3_f32 %5, %5, %8, %9\n\t"
            "v_med3_f32 %6, %6, %8, %9\n\t"
            "v_med3_f32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CVT_UBYTE) {
            asm volatile(REP16(
            "v_cvt_f32_ubyte1 %0, %8\n\t"
            "v_cvt_f32_ubyte1 %1, %8\n\t"
            "v_cvt_f32_ubyte1 %2, %8\n\t"
            "v_cvt_f32_ubyte1 %3, %8\n\t"
            "v_cvt_f32_ubyte1 %4, %8\n\t"
            "v_cvt_f32_ubyte1 %5, %8\n\t"
            "v_cvt_f32_ubyte1 %6, %8\n\t"
            "v_cvt_f32_ubyte1 %7, %8\n\t"
            )
              : ACC8 : "v"(u), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CVT_F32_U32) {
            asm volatile(REP16(
            "v_cvt_f32_u32 %0, %0\n\t"
            "v_cvt_f32_u32 %1, %1\n\t"
            "v_cvt_f32_u32 %2, %2\n\t"
            "v_cvt_f32_u32 %3, %3\n\t"
            "v_cvt_f32_u32 %4, %4\n\t"
            "v_cvt_f32_u32 %5, %5\n\t"
            "v_cvt_f32_u32 %6, %6\n\t"
            "v_cvt_f32_u32 %7, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CVT_U32_F32) {
            asm volatile(REP16(
            "v_cvt_u32_f32 %0, %0\n\t"
            "v_cvt_u32_f32 %1, %1\n\t"
            "v_cvt_u32_f32 %2, %2\n\t"
            "v_cvt_u32_f32 %3, %3\n\t"
            "v_cvt_u32_f32 %4, %4\n\t"
            "v_cvt_u32_f32 %5, %5\n\t"
            "v_cvt_u32_f32 %6, %6\n\t"
            "v_cvt_u32_f32 %7, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CVT_F32_F16) {
            asm volatile(REP16(
            "v_cvt_f32_f16 %0, %8\n\t"
            "v_cvt_f32_f16 %1, %8\n\t"
            "v_cvt_f32_f16 %2, %8\n\t"
            "v_cvt_f32_f16 %3, %8\n\t"
            "v_cvt_f32_f16 %4, %8\n\t"
            "v_cvt_f32_f16 %5, %8\n\t"
            "v_cvt_f32_f16 %6, %8\n\t"
            "v_cvt_f32_f16 %7, %8\n\t"
            )
              : ACC8 : "v"(u), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CVT_F16_F32) {
            asm volatile(REP16(
            "v_cvt_f16_f32 %0, %8\n\t"
            "v_cvt_f16_f32 %1, %8\n\t"
            "v_cvt_f16_f32 %2, %8\n\t"
            "v_cvt_f16_f32 %3, %8\n\t"
            "v_cvt_f16_f32 %4, %8\n\t"
            "v_cvt_f16_f32 %5, %8\n\t"
            "v_cvt_f16_f32 %6, %8\n\t"
            "v_cvt_f16_f32 %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == FLOOR) {
            asm volatile(REP16(
            "v_floor_f32 %0, %0\n\t"
            "v_floor_f32 %1, %1\n\t"
            "v_floor_f32 %2, %2\n\t"
            "v_floor_f32 %3, %3\n\t"
            "v_floor_f32 %4, %4\n\t"
            "v_floor_f32 %5, %5\n\t"
            "v_floor_f32 %6, %6\n\t"
            "v_floor_f32 %7, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == LDEXP) {
            asm volatile(REP16(
            "v_ldexp_f32 %0, %0, 1\n\t"
            "v_ldexp_f32 %1, %1, 1\n\t"
            "v_ldexp_f32 %2, %2, 1\n\t"
            "v_ldexp_f32 %3, %3, 1\n\t"
            "v_ldexp_f32 %4, %4, 1\n\t"
            "v_ldexp_f32 %5, %5, 1\n\t"
            "v_ldexp_f32 %6, %6, 1\n\t"
            "v_ldexp_f32 %7, %7, 1\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == RCP) {
            asm volatile(REP16(
            "v_rcp_f32 %0, %0\n\t"
            "v_rcp_f32 %1, %1\n\t"
            "v_rcp_f32 %2, %2\n\t"
            "v_rcp_f32 %3, %3\n\t"
            "v_rcp_f32 %4, %4\n\t"
            "v_rcp_f32 %5, %5\n\t"
            "v_rcp_f32 %6, %6\n\t"
            "v_rcp_f32 %7, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == SQRT) {
            asm volatile(REP16(
            "v_sqrt_f32 %0, %0\n\t"
            "v_sqrt_f32 %1, %1\n\t"
            "v_sqrt_f32 %2, %2\n\t"
            "v_sqrt_f32 %3, %3\n\t"
            "v_sqrt_f32 %4, %4\n\t"
            "v_sqrt_f32 %5, %5\n\t"
            "v_sqrt_f32 %6, %6\n\t"
            "v_sqrt_f32 %7, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == RSQ) {
            asm volatile(REP16(
            "v_rsq_f32 %0, %0\n\t"
            "v_rsq_f32 %1, %1\n\t"
            "v_rsq_f32 %2, %2\n\t"
            "v_rsq_f32 %3, %3\n\t"
            "v_rsq_f32 %4, %4\n\t"
            "v_rsq_f32 %5, %5\n\t"
            "v_rsq_f32 %6, %6\n\t"
            "v_rsq_f32 %7, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == DIV_SCALE) {
            asm volatile(REP16(
            "v_div_scale_f32 %0, vcc, %0, %8, %0\n\t"
            "v_div_scale_f32 %1, vcc, %1, %8, %1\n\t"
            "v_div_scale_f32 %2, vcc, %2, %8, %2\n\t"
            "v_div_scale_f32 %3, vcc, %3, %8, %3\n\t"
            "v_div_scale_f32 %4, vcc, %4, %8, %4\n\t"
            "v_div_scale_f32 %5, vcc, %5, %8, %5\n\t"
            "v_div_scale_f32 %6, vcc, %6, %8, %6\n\t"
            "v_div_scale_f32 %7, vcc, %7, %8, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == DIV_FMAS) {
            asm volatile(
            "v_cmp_lt_f32_e32 vcc, %8, %9\n\ts_nop 4\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
            "v_div_fmas_f32 %0, %0, %8, %9\n\t"
            "v_div_fmas_f32 %1, %1, %8, %9\n\t"
            "v_div_fmas_f32 %2, %2, %8, %9\n\t"
            "v_div_fmas_f32 %3, %3, %8, %9\n\t"
            "v_div_fmas_f32 %4, %4, %8, %9\n\t"
            "v_div_fmas_f32 %5, %5, %8, %9\n\t"
            "v_div_fmas_f32 %6, %6, %8, %9\n\t"
            "v_div_fmas_f32 %7, %7, %8, %9\n\t"
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == DIV_FIXUP) {
            asm volatile(REP16(
            "v_div_fixup_f32 %0, %0, %8, %9\n\t"
            "v_div_fixup_f32 %1, %1, %8, %9\n\t"
            "v_div_fixup_f32 %2, %2, %8, %9\n\t"
            "v_div_fixup_f32 %3, %3, %8, %9\n\t"
            "v_div_fixup_f32 %4, %4, %8, %9\n\t"
            "v_div_fixup_f32 %5, %5, %8, %9\n\t"
            "v_div_fixup_f32 %6, %6, %8, %9\n\t"
            "v_div_fixup_f32 %7, %7, %8, %9\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CMP_E64) {
            asm volatile(REP16(
            "v_cmp_lt_f32_e64 s[20:21], %0, %8\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %1, %8\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %2, %8\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %3, %8\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %4, %8\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %5, %8\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %6, %8\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CMP_E32) {
            asm volatile(REP16(
            "v_cmp_lt_f32_e32 vcc, %0, %8\n\t"
            "v_cmp_lt_f32_e32 vcc, %1, %8\n\t"
            "v_cmp_lt_f32_e32 vcc, %2, %8\n\t"
            "v_cmp_lt_f32_e32 vcc, %3, %8\n\t"
            "v_cmp_lt_f32_e32 vcc, %4, %8\n\t"
            "v_cmp_lt_f32_e32 vcc, %5, %8\n\t"
            "v_cmp_lt_f32_e32 vcc, %6, %8\n\t"
            "v_cmp_lt_f32_e32 vcc, %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CMP_U32_E64) {
            asm volatile(REP16(
            "v_cmp_lt_u32_e64 s[20:21], %0, %8\n\t"
            "v_cmp_lt_u32_e64 s[20:21], %1, %8\n\t"
            "v_cmp_lt_u32_e64 s[20:21], %2, %8\n\t"
            "v_cmp_lt_u32_e64 s[20:21], %3, %8\n\t"
            "v_cmp_lt_u32_e64 s[20:21], %4, %8\n\t"
            "v_cmp_lt_u32_e64 s[20:21], %5, %8\n\t"
            "v_cmp_lt_u32_e64 s[20:21], %6, %8\n\t"
            "v_cmp_lt_u32_e64 s[20:21], %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CMPX) {
            asm volatile(REP128(
            "v_cmpx_le_f32_e64 s[20:21], %8, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CNDMASK_E64) {
            asm volatile(
            "v_cmp_lt_f32_e64 s[20:21], %8, %9\n\ts_nop 4\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CNDMASK_E64_VCC) {
            asm volatile(
            "v_cmp_lt_f32_e32 vcc, %8, %9\n\ts_nop 4\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e64 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e64 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e64 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e64 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e64 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e64 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e64 %7, %7, %8, vcc\n\t"
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CNDMASK_E32) {
            asm volatile(
            "v_cmp_lt_f32_e32 vcc, %8, %9\n\ts_nop 4\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == PAIR_E32) {
            asm volatile(REP16(
            "v_cmp_lt_f32_e32 vcc, %0, %8\n\tv_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %1, %8\n\tv_cndmask_b32_e32 %1, %1, %8, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %2, %8\n\tv_cndmask_b32_e32 %2, %2, %8, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %3, %8\n\tv_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %4, %8\n\tv_cndmask_b32_e32 %4, %4, %8, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %5, %8\n\tv_cndmask_b32_e32 %5, %5, %8, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %6, %8\n\tv_cndmask_b32_e32 %6, %6, %8, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %7, %8\n\tv_cndmask_b32_e32 %7, %7, %8, vcc\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == PAIR_E64) {
            asm volatile(REP16(
            "v_cmp_lt_f32_e64 s[20:21], %0, %8\n\tv_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %1, %8\n\tv_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %2, %8\n\tv_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %3, %8\n\tv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %4, %8\n\tv_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %5, %8\n\tv_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %6, %8\n\tv_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %7, %8\n\tv_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CMP_4SEL_E64) {
            asm volatile(REP4(
            "v_cmp_lt_f32_e64 s[20:21], %0, %8\n\tv_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %9, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %9, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %1, %8\n\tv_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %9, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %9, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %2, %8\n\tv_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %9, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\tv_cndmask_b32_e64 %5, %5, %9, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %3, %8\n\tv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %9, s[20:21]\n\tv_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\tv_cndmask_b32_e64 %6, %6, %9, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %4, %8\n\tv_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\tv_cndmask_b32_e64 %5, %5, %9, s[20:21]\n\tv_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\tv_cndmask_b32_e64 %7, %7, %9, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %5, %8\n\tv_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\tv_cndmask_b32_e64 %6, %6, %9, s[20:21]\n\tv_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %9, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %6, %8\n\tv_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\tv_cndmask_b32_e64 %7, %7, %9, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %9, s[20:21]\n\t"
            "v_cmp_lt_f32_e64 s[20:21], %7, %8\n\tv_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %9, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %9, s[20:21]\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "s20", "s21");
        }
        else if (OP == CMP_4SEL_VCC) {
            asm volatile(REP4(
            "v_cmp_lt_f32_e32 vcc, %0, %8\n\tv_cndmask_b32_e32 %0, %0, %8, vcc\n\tv_cndmask_b32_e32 %1, %1, %9, vcc\n\tv_cndmask_b32_e32 %2, %2, %8, vcc\n\tv_cndmask_b32_e32 %3, %3, %9, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %1, %8\n\tv_cndmask_b32_e32 %1, %1, %8, vcc\n\tv_cndmask_b32_e32 %2, %2, %9, vcc\n\tv_cndmask_b32_e32 %3, %3, %8, vcc\n\tv_cndmask_b32_e32 %4, %4, %9, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %2, %8\n\tv_cndmask_b32_e32 %2, %2, %8, vcc\n\tv_cndmask_b32_e32 %3, %3, %9, vcc\n\tv_cndmask_b32_e32 %4, %4, %8, vcc\n\tv_cndmask_b32_e32 %5, %5, %9, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %3, %8\n\tv_cndmask_b32_e32 %3, %3, %8, vcc\n\tv_cndmask_b32_e32 %4, %4, %9, vcc\n\tv_cndmask_b32_e32 %5, %5, %8, vcc\n\tv_cndmask_b32_e32 %6, %6, %9, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %4, %8\n\tv_cndmask_b32_e32 %4, %4, %8, vcc\n\tv_cndmask_b32_e32 %5, %5, %9, vcc\n\tv_cndmask_b32_e32 %6, %6, %8, vcc\n\tv_cndmask_b32_e32 %7, %7, %9, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %5, %8\n\tv_cndmask_b32_e32 %5, %5, %8, vcc\n\tv_cndmask_b32_e32 %6, %6, %9, vcc\n\tv_cndmask_b32_e32 %7, %7, %8, vcc\n\tv_cndmask_b32_e32 %0, %0, %9, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %6, %8\n\tv_cndmask_b32_e32 %6, %6, %8, vcc\n\tv_cndmask_b32_e32 %7, %7, %9, vcc\n\tv_cndmask_b32_e32 %0, %0, %8, vcc\n\tv_cndmask_b32_e32 %1, %1, %9, vcc\n\t"
            "v_cmp_lt_f32_e32 vcc, %7, %8\n\tv_cndmask_b32_e32 %7, %7, %8, vcc\n\tv_cndmask_b32_e32 %0, %0, %9, vcc\n\tv_cndmask_b32_e32 %1, %1, %8, vcc\n\tv_cndmask_b32_e32 %2, %2, %9, vcc\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc");
        }
        else if (OP == EXEC_MOV) {
            asm volatile("s_mov_b64 s[22:23], exec\n\tv_cmp_lt_f32_e64 s[20:21], %8, %9\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %9\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %9\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %9\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %9\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %9\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %9\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %9\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
            "s_mov_b64 exec, s[20:21]\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %9\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\ts_mov_b64 exec, s[22:23]\n\t"
              : ACC8 : "v"(b), "v"(c) : "s20", "s21", "s22", "s23");
        }
        else if (OP == SWAP) {
            asm volatile(REP16(
            "v_swap_b32 %0, %1\n\t"
            "v_swap_b32 %1, %2\n\t"
            "v_swap_b32 %2, %3\n\t"
            "v_swap_b32 %3, %4\n\t"
            "v_swap_b32 %4, %5\n\t"
            "v_swap_b32 %5, %6\n\t"
            "v_swap_b32 %6, %7\n\t"
            "v_swap_b32 %7, %0\n\t"
            )
              : ACC8 : "v"(b), "v"(c));
        }
        else if (OP == READFIRSTLANE) {
            asm volatile(REP16(
            "v_readfirstlane_b32 s22, %0\n\t"
            "v_readfirstlane_b32 s22, %1\n\t"
            "v_readfirstlane_b32 s22, %2\n\t"
            "v_readfirstlane_b32 s22, %3\n\t"
            "v_readfirstlane_b32 s22, %4\n\t"
            "v_readfirstlane_b32 s22, %5\n\t"
            "v_readfirstlane_b32 s22, %6\n\t"
            "v_readfirstlane_b32 s22, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MBCNT) {
            asm volatile(REP16(
            "v_mbcnt_lo_u32_b32 %0, -1, %0\n\t"
            "v_mbcnt_lo_u32_b32 %1, -1, %1\n\t"
            "v_mbcnt_lo_u32_b32 %2, -1, %2\n\t"
            "v_mbcnt_lo_u32_b32 %3, -1, %3\n\t"
            "v_mbcnt_lo_u32_b32 %4, -1, %4\n\t"
            "v_mbcnt_lo_u32_b32 %5, -1, %5\n\t"
            "v_mbcnt_lo_u32_b32 %6, -1, %6\n\t"
            "v_mbcnt_lo_u32_b32 %7, -1, %7\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MOV_DPP) {
            asm volatile(REP16(
            "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MOV_SDWA) {
            asm volatile(REP16(
            "v_mov_b32_sdwa %0, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\t"
            "v_mov_b32_sdwa %1, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\t"
            "v_mov_b32_sdwa %2, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\t"
            "v_mov_b32_sdwa %3, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\t"
            "v_mov_b32_sdwa %4, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\t"
            "v_mov_b32_sdwa %5, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\t"
            "v_mov_b32_sdwa %6, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\t"
            "v_mov_b32_sdwa %7, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == OR_SDWA) {
            asm volatile(REP16(
            "v_or_b32_sdwa %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_or_b32_sdwa %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_or_b32_sdwa %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_or_b32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_or_b32_sdwa %4, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_or_b32_sdwa %5, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_or_b32_sdwa %6, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_or_b32_sdwa %7, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == ADD_F32_SDWA) {
            asm volatile(REP16(
            "v_add_f32_sdwa %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_add_f32_sdwa %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_add_f32_sdwa %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_add_f32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_add_f32_sdwa %4, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_add_f32_sdwa %5, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_add_f32_sdwa %6, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_add_f32_sdwa %7, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CVT_F32_F16_SDWA) {
            asm volatile(REP16(
            "v_cvt_f32_f16_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_cvt_f32_f16_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_cvt_f32_f16_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_cvt_f32_f16_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_cvt_f32_f16_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_cvt_f32_f16_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_cvt_f32_f16_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_cvt_f32_f16_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == CVT_UBYTE3) {
            asm volatile(REP16(
            "v_cvt_f32_ubyte3 %0, %8\n\t"
            "v_cvt_f32_ubyte3 %1, %8\n\t"
            "v_cvt_f32_ubyte3 %2, %8\n\t"
            "v_cvt_f32_ubyte3 %3, %8\n\t"
            "v_cvt_f32_ubyte3 %4, %8\n\t"
            "v_cvt_f32_ubyte3 %5, %8\n\t"
            "v_cvt_f32_ubyte3 %6, %8\n\t"
            "v_cvt_f32_ubyte3 %7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MUL_F32_ABS) {
            asm volatile(REP16(
            "v_mul_f32_e64 %0, |%0|, %8\n\t"
            "v_mul_f32_e64 %1, |%1|, %8\n\t"
            "v_mul_f32_e64 %2, |%2|, %8\n\t"
            "v_mul_f32_e64 %3, |%3|, %8\n\t"
            "v_mul_f32_e64 %4, |%4|, %8\n\t"
            "v_mul_f32_e64 %5, |%5|, %8\n\t"
            "v_mul_f32_e64 %6, |%6|, %8\n\t"
            "v_mul_f32_e64 %7, |%7|, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MAX_F32_E64) {
            asm volatile(REP16(
            "v_max_f32_e64 %0, -%0, %8\n\t"
            "v_max_f32_e64 %1, -%1, %8\n\t"
            "v_max_f32_e64 %2, -%2, %8\n\t"
            "v_max_f32_e64 %3, -%3, %8\n\t"
            "v_max_f32_e64 %4, -%4, %8\n\t"
            "v_max_f32_e64 %5, -%5, %8\n\t"
            "v_max_f32_e64 %6, -%6, %8\n\t"
            "v_max_f32_e64 %7, -%7, %8\n\t"
            )
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == LSHL_ADD_U64) {
            asm volatile(REP16(
            "v_lshl_add_u64 %0, %0, 3, %8\n\t"
            "v_lshl_add_u64 %1, %1, 3, %8\n\t"
            "v_lshl_add_u64 %2, %2, 3, %8\n\t"
            "v_lshl_add_u64 %3, %3, 3, %8\n\t"
            "v_lshl_add_u64 %4, %4, 3, %8\n\t"
            "v_lshl_add_u64 %5, %5, 3, %8\n\t"
            "v_lshl_add_u64 %6, %6, 3, %8\n\t"
            "v_lshl_add_u64 %7, %7, 3, %8\n\t"
            )
              : PACC8 : "v"(pb), "v"(pc) : "vcc", "s20", "s21", "s22");
        }
        else if (OP == MIX_TRAV) {
            asm volatile(
            #include "valu_mix_body_trav.inc"
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23");
        }
        else if (OP == MIX_MEGA) {
            asm volatile(
            #include "valu_mix_body_mega.inc"
              : ACC8 : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23");
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
static const double kInstrPerIter[N_OPS] = {128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 256.0, 256.0, 160.0, 160.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, 128.0, kMixCount_trav, kMixCount_mega};
static kern_t kKernels[N_OPS] = {k_calib<FMA>, k_calib<MUL>, k_calib<ADD_F32>, k_calib<SUB_F32>, k_calib<FMAC>, k_calib<MOV>, k_calib<AND>, k_calib<OR>, k_calib<XOR>, k_calib<ADD_U32>, k_calib<SUB_U32>, k_calib<MUL_U24>, k_calib<MAD_U24>, k_calib<FMA_DEP>, k_calib<FMA_HALF_EXEC>, k_calib<FMA_MIX>, k_calib<FMA_MIX_HI>, k_calib<PK_FMA>, k_calib<PK_MUL>, k_calib<PK_ADD>, k_calib<PK_FMA_F16>, k_calib<PK_MAX_F16>, k_calib<LSHL>, k_calib<LSHR>, k_calib<ASHR>, k_calib<LSHL_OR>, k_calib<LSHL_ADD>, k_calib<AND_OR>, k_calib<ADD3>, k_calib<BFE>, k_calib<BFI>, k_calib<PERM>, k_calib<ALIGNBIT>, k_calib<MUL_LO>, k_calib<MIN_F32>, k_calib<MAX_F32>, k_calib<MIN_U32>, k_calib<MAX3>, k_calib<MED3>, k_calib<CVT_UBYTE>, k_calib<CVT_F32_U32>, k_calib<CVT_U32_F32>, k_calib<CVT_F32_F16>, k_calib<CVT_F16_F32>, k_calib<FLOOR>, k_calib<LDEXP>, k_calib<RCP>, k_calib<SQRT>, k_calib<RSQ>, k_calib<DIV_SCALE>, k_calib<DIV_FMAS>, k_calib<DIV_FIXUP>, k_calib<CMP_E64>, k_calib<CMP_E32>, k_calib<CMP_U32_E64>, k_calib<CMPX>, k_calib<CNDMASK_E64>, k_calib<CNDMASK_E64_VCC>, k_calib<CNDMASK_E32>, k_calib<PAIR_E32>, k_calib<PAIR_E64>, k_calib<CMP_4SEL_E64>, k_calib<CMP_4SEL_VCC>, k_calib<EXEC_MOV>, k_calib<SWAP>, k_calib<READFIRSTLANE>, k_calib<MBCNT>, k_calib<MOV_DPP>, k_calib<MOV_SDWA>, k_calib<OR_SDWA>, k_calib<ADD_F32_SDWA>, k_calib<CVT_F32_F16_SDWA>, k_calib<CVT_UBYTE3>, k_calib<MUL_F32_ABS>, k_calib<MAX_F32_E64>, k_calib<LSHL_ADD_U64>, k_calib<MIX_TRAV>, k_calib<MIX_MEGA>};

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
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iters\": %u, \"mix_trav\": \"%s\", \"mix_mega\": \"%s\", \"results\": [\n", prop.gcnArchName, n_cu,
                prop.clockRate / 1000, iters, kMixHist_trav, kMixHist_mega);
    bool first = true;
    const int wps_list[] = {1, 2, 3, 4, 6, 8};
    for (int op = 0; op < N_OPS; ++op) {
        for (int wps : wps_list) {
            // W waves per SIMD: blocks of 256*w_blk threads, n_blk per CU; LDS keeps n_blk blocks per CU and no more
            const int n_blk = wps > 4 ? 2 : 1, w_blk = wps / n_blk; // 6 -> 2 x 3, 8 -> 2 x 4
            const size_t lds = n_blk == 1 ? 96 * 1024 : 64 * 1024;
            CHK(hipFuncSetAttribute((const void*)kKernels[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const dim3 grid(n_cu * n_blk), block(256 * w_blk);
            const double ipi = kInstrPerIter[op];
            const uint32_t it = std::max<uint32_t>(8u, (uint32_t)((double)iters * 128.0 / ipi) / ((op == RCP || op == SQRT || op == RSQ) ? 2u : 1u));
            hipLaunchKernelGGL(kKernels[op], grid, block, lds, 0, it / 8 + 1, 1.0f, d_st, d_sink); // warm-up
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kKernels[op], grid, block, lds, 0, it, 1.0f, d_st, d_sink);
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
            const double per_wave = (double)it * ipi;
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
