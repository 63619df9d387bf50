"""Static VALU cycle estimate of the kernels in a device assembly file, from the calibrated per-opcode issue costs
(profiles/r02_valu_calibration.json): python3 scripts/isa_cost.py build/rt_dev.s [kernel-substring ...]"""
import collections, re, sys
TWO = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32",
       "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mac_f32", "v_not_b32"}
EIGHT = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_swap_b32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}


def cost(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op == "v_cndmask_b32_e32":
        return 8  # VCC form: 4 right after its compare, 16 otherwise
    if base in TWO and not op.endswith(("_sdwa", "_dpp")):
        return 2
    if base in EIGHT:
        return 8
    if base.startswith("v_pk_") or base in ("v_mov_b64", "v_lshl_add_u64", "v_lshlrev_b64"):
        return 4
    return 4


src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2:]
name, rows = None, collections.OrderedDict()
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1)
        rows[name] = collections.Counter()
        continue
    if name is None:
        continue
    t = ln.split()
    if t and t[0].startswith(("v_", "s_", "ds_", "global_", "scratch_", "buffer_", "flat_")):
        rows[name][t[0]] += 1
for k, h in rows.items():
    if want and not any(w in k for w in want):
        continue
    valu = {o: n for o, n in h.items() if o.startswith("v_")}
    cyc = sum(cost(o) * n for o, n in valu.items())
    print(f"{k[:70]}: VALU {sum(valu.values())} instr, ~{cyc} issue cycles ({cyc / max(1, sum(valu.values())):.2f}/instr) | SALU {sum(n for o, n in h.items() if o.startswith('s_'))}"
          f" | LDS {sum(n for o, n in h.items() if o.startswith('ds_'))} | VMEM {sum(n for o, n in h.items() if o.startswith(('global_', 'scratch_', 'buffer_', 'flat_')))}"
          f" | cndmask e32 {h.get('v_cndmask_b32_e32', 0)} e64 {h.get('v_cndmask_b32_e64', 0)}")
