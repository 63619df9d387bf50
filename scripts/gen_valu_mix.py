"""Generates scripts/valu_mix_body_{trav,mega}.inc for scripts/valu_calib.hip: an inline-asm body that replays the STATIC VALU opcode mix of a
render kernel (k_wf_extend<false> = the traversal loop, k_megakernel<false> = traversal + shading) on independent registers, so that the
microbenchmark measures the VALU issue ceiling FOR THAT MIX (cycles per instruction and SIMD with N waves resident).
   python3 scripts/gen_valu_mix.py sycl-ray-tracer_amd/csrc/build/rt_dev.s
(rt_dev.s: hipcc <the Makefile's HIPFLAGS> --cuda-device-only -S rt_mi355x.hip). Operands are generic: what is reproduced is the opcode
histogram (2-cycle vs 4-cycle vs transcendental classes), not the data flow. Packed / 64-bit forms are replayed as two 32-bit instructions
of the same class."""
import collections, re, sys
from pathlib import Path

src = Path(sys.argv[1]).read_text().splitlines()
HERE = Path(__file__).resolve().parent


def body_of(mangled_prefix):
    out, on = [], False
    for ln in src:
        if ln.startswith(mangled_prefix) and ln.rstrip().endswith(":") or (ln.startswith(mangled_prefix) and ":" in ln.split()[0]):
            on = True
            continue
        if on:
            if "s_endpgm" in ln:
                break
            out.append(ln)
    return out


VOP1 = {"v_mov_b32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_cvt_f32_f16", "v_cvt_f16_f32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_cvt_i32_f32",
        "v_cvt_f32_i32", "v_floor_f32", "v_fract_f32", "v_rndne_f32", "v_bfrev_b32", "v_not_b32", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1",
        "v_cvt_f32_ubyte2", "v_cvt_f32_ubyte3", "v_frexp_mant_f32", "v_frexp_exp_i32_f32", "v_trunc_f32", "v_ffbh_u32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32", "v_ffbl_b32", "v_ceil_f32"}
SHIFT2 = {"v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32"}
VOP3_3 = {"v_fma_f32", "v_max3_f32", "v_min3_f32", "v_med3_f32", "v_lshl_add_u32", "v_lshl_or_b32", "v_and_or_b32", "v_bfe_u32", "v_bfi_b32",
          "v_perm_b32", "v_add3_u32", "v_mad_u32_u24", "v_alignbit_b32", "v_or3_b32", "v_xad_u32", "v_add_lshl_u32", "v_div_fixup_f32",
          "v_div_fmas_f32", "v_mad_u64_u32", "v_mad_i64_i32", "v_bfe_i32", "v_mad_i32_i24", "v_med3_i32", "v_med3_u32", "v_max3_u32", "v_min3_u32", "v_max3_i32", "v_min3_i32"}
PACKED = {"v_pk_mul_f32": "v_mul_f32", "v_pk_add_f32": "v_add_f32", "v_pk_fma_f32": "v_fma_f32", "v_mov_b64": "v_mov_b32",
          "v_lshlrev_b64": "v_lshlrev_b32", "v_lshl_add_u64": "v_lshl_add_u32", "v_pk_mov_b32": "v_mov_b32"}


def emit(op, k):
    """one asm line for opcode `op` on accumulator %k (b = %8, c = %9); None = not a VALU instruction we replay"""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in PACKED:
        one = emit(PACKED[base], k)
        return one + one
    a = f"%{k}"
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        b32 = base.replace("v_cmpx", "v_cmp")
        if "_u64" in b32 or "_i64" in b32 or "_f64" in b32:
            b32 = "v_cmp_lt_u32"
        if op.endswith("_e64"):
            return f'"{b32}_e64 s[20:21], {a}, %8\\n\\t"'
        return f'"{b32}_e32 vcc, {a}, %8\\n\\t"'
    if base == "v_cndmask_b32":
        return f'"v_cndmask_b32_e64 {a}, {a}, %8, s[20:21]\\n\\t"' if op.endswith("_e64") else f'"v_cndmask_b32_e32 {a}, {a}, %8, vcc\\n\\t"'
    if base in VOP1:
        return f'"{base} {a}, {a}\\n\\t"'
    if base in SHIFT2:
        return f'"{base} {a}, 1, {a}\\n\\t"'
    if base in ("v_div_scale_f32",):
        return f'"v_div_scale_f32 {a}, vcc, {a}, %8, {a}\\n\\t"'
    if base in ("v_div_fmas_f32",):
        return f'"v_div_fmas_f32 {a}, {a}, %8, %9\\n\\t"'
    if base in VOP3_3:
        if base in ("v_lshl_add_u32", "v_lshl_or_b32"):
            return f'"{base} {a}, {a}, 2, %8\\n\\t"'
        if base in ("v_bfe_u32", "v_bfe_i32"):
            return f'"{base} {a}, {a}, 8, 8\\n\\t"'
        if base in ("v_mad_u64_u32", "v_mad_i64_i32"):
            return f'"v_mad_u32_u24 {a}, {a}, %8, %9\\n\\t"'
        return f'"{base} {a}, {a}, %8, %9\\n\\t"'
    if base in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):
        return f'"{base} {a}, -1, {a}\\n\\t"'
    if base == "v_readlane_b32":
        return f'"v_readlane_b32 s22, {a}, 3\\n\\t"'
    if base == "v_writelane_b32":
        return f'"v_writelane_b32 {a}, s23, 3\\n\\t"'
    if base == "v_readfirstlane_b32":
        return f'"v_readfirstlane_b32 s22, {a}\\n\\t"'
    if base in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32"):
        return f'"v_add_u32 {a}, {a}, %8\\n\\t"'
    if base == "v_bitop3_b32":
        return f'"v_bitop3_b32 {a}, {a}, %8, %9 bitop3:0x36\\n\\t"'
    if base == "v_fmac_f32":
        return f'"v_fmac_f32 {a}, %8, %9\\n\\t"'
    if base == "v_ldexp_f32":
        return f'"v_ldexp_f32 {a}, {a}, 1\\n\\t"'
    if base == "v_accvgpr_write_b32" or base == "v_accvgpr_read_b32" or base == "v_nop":
        return f'"v_mov_b32 {a}, {a}\\n\\t"'
    if base.startswith("v_"):
        return f'"{base} {a}, {a}, %8\\n\\t"'  # VOP2 default
    return None


def generate(name, prefix):
    lines = body_of(prefix)
    hist = collections.Counter()
    for ln in lines:
        t = ln.split()
        if not t or t[0].startswith((";", ".", "//")) or t[0].endswith(":"):
            continue
        if t[0].startswith("v_"):
            hist[t[0]] += 1
    total = sum(hist.values())
    # interleave opcodes proportionally (largest-remainder round robin) so that no long run of one class forms
    seq, credit = [], collections.Counter()
    for _ in range(total):
        for op, n in hist.items():
            credit[op] += n
        op = max(credit, key=lambda o: credit[o])
        credit[op] -= total
        seq.append(op)
    out, count, skipped = [], 0, collections.Counter()
    for i, op in enumerate(seq):
        e = emit(op, i % 8)
        if e is None:
            skipped[op] += 1
            continue
        out.append(e)
        count += e.count("\\n\\t")
    (HERE / f"valu_mix_body_{name}.inc").write_text("// generated by scripts/gen_valu_mix.py from " + prefix + "\n" + "\n".join(out) + "\n")
    summary = ", ".join(f"{op} {n}" for op, n in hist.most_common())
    return count, total, summary, skipped


res = {n: generate(n, p) for n, p in (("trav", "_ZN2rt11k_wf_extendILb0"), ("mega", "_ZN2rt12k_megakernelILb0"))}
(HERE / "valu_mix_count.inc").write_text("// generated by scripts/gen_valu_mix.py\n" +
                                         "".join(f"static const int kMixCount_{n} = {c};\n" for n, (c, _, _, _) in res.items()) +
                                         "".join(f"static const char* kMixHist_{n} = \"{s}\";\n" for n, (_, _, s, _) in res.items()))
for n, (c, t, s, sk) in res.items():
    print(n, "VALU instructions in the kernel:", t, "replayed:", c, "skipped:", dict(sk))
    print("   ", s[:600])
