"""profiles/r02_valu_calibration.json from the outputs of scripts/valu_calib.sh (gpurun_out/r02/valu_calib*.json + the two PMC passes):
the measured VALU issue cost per instruction kind (cycles per wave64 instruction and SIMD, by resident waves per SIMD), the cost of the
render kernels' own static opcode mixes, and what the SQ counters report on a known instruction count (the calibration of
valu_issue_busy / lane_utilisation in profiles/rNN_valu.json).     python3 scripts/valu_calib_summary.py > profiles/r02_valu_calibration.json"""
import os, collections, csv, glob, json, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
R = REPO / "gpurun_out" / "r02"
plain = json.loads((R / "valu_calib.json").read_text())
ops = collections.OrderedDict()
for r in plain["results"]:
    ops.setdefault(r["op"], {})[r["waves_per_simd"]] = r
table = {op: {f"w{w}": round(v[w]["cycles_per_instr_per_simd"], 2) for w in sorted(v)} for op, v in ops.items()}
classes = {"2-cycle (full rate)": [], "4-cycle (half rate)": [], "8-cycle (transcendental, v_swap)": [], "other": []}
for op, v in ops.items():
    if op.startswith(("pair", "group", "static")):
        continue
    c = v[6]["cycles_per_instr_per_simd"]
    key = "2-cycle (full rate)" if c < 3.0 else "4-cycle (half rate)" if c < 5.0 else "8-cycle (transcendental, v_swap)" if c < 10 else "other"
    classes[key].append(op)


def pmc(dirname, under):
    """per timed dispatch (every second one: warm-up first) of the run under rocprofv3"""
    res = json.loads((R / under).read_text())["results"]
    f = glob.glob(str(R / dirname / "**" / "*counter_collection.csv"), recursive=True)
    if not f:
        return {}
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):  # gpurun_out keeps earlier runs: the newest
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    out = {}
    for i, r in enumerate(res):
        if 2 * i + 1 < len(ids):
            out[(r["op"], r["waves_per_simd"])] = (r, rows[ids[2 * i + 1]])
    return out


p1, p2 = pmc("valu_calib_pmc", "valu_calib_under_pmc.json"), pmc("valu_calib_pmc2", "valu_calib_under_pmc2.json")
counters = {}
for key in (("v_fma_f32", 6), ("v_cvt_f32_ubyte1", 6), ("v_rcp_f32", 6), ("v_fma_f32 (lanes 32-63 masked off)", 6),
            ("static VALU mix of k_wf_extend (traversal loop)", 6), ("static VALU mix of k_megakernel", 6)):
    if key not in p1:
        continue
    r, c = p1[key]
    n = r["valu_instructions"]
    e = {"issued_wave_instructions": n, "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": c.get("SQ_ACTIVE_INST_VALU"),
         "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE"), "SQ_BUSY_CYCLES": c.get("SQ_BUSY_CYCLES"),
         "kernel_cycles_from_GRBM_GUI_ACTIVE_div_8": c.get("GRBM_GUI_ACTIVE", 0) / 8,
         "cycles_per_instr_per_simd_from_counters": c.get("GRBM_GUI_ACTIVE", 0) / 8 / (c.get("SQ_INSTS_VALU", 1) / 1024)}
    if key in p2:
        c2 = p2[key][1]
        e["SQ_THREAD_CYCLES_VALU_per_instruction"] = c2.get("SQ_THREAD_CYCLES_VALU", 0) / n
        e["SQ_WAVE_CYCLES"] = c2.get("SQ_WAVE_CYCLES")
        e["SQ_WAIT_INST_ANY"] = c2.get("SQ_WAIT_INST_ANY")
    counters[f"{key[0]} @ {key[1]} waves/SIMD"] = e
mix_mega = ops["static VALU mix of k_megakernel"][6]["cycles_per_instr_per_simd"]
mix_trav = ops["static VALU mix of k_wf_extend (traversal loop)"][6]["cycles_per_instr_per_simd"]
out = {
    "source": "scripts/valu_calib.hip via scripts/valu_calib.sh on MI355X (gfx950), round 2: every wave issues N independent instructions of one kind "
              "(one inline-asm block of 128 per loop iteration), W waves per SIMD forced by LDS; cycles = loop span on the 100 MHz s_memrealtime clock x "
              "the shader clock measured in the same loop (s_memtime / s_memrealtime).",
    "findings": [
        "gfx950 issues a wave64 VALU instruction in 2 cycles ONLY for the plain fp32 / integer VOP2 class and v_fma_f32 (v_fma/mul/add/sub/fmac_f32, v_mov_b32, "
        "v_and/or/xor_b32, v_add/sub_u32, v_lshrrev_b32, v_ashrrev_i32) and only with >= 2 waves resident per SIMD (one wave alone: ~5 cycles for everything).",
        "Everything else the traversal is made of is HALF rate, 4 cycles: v_cvt_f32_ubyteN, v_min/max(3)_f32, every v_cmp, v_cndmask_b32_e64, v_lshlrev_b32, "
        "v_lshl_or/add, v_bfe/bfi/perm, v_fma_mix_f32, all SDWA / DPP forms, all packed (v_pk_*) forms, v_readfirstlane, v_mbcnt. v_rcp/v_sqrt/v_rsq and v_swap_b32: 8.",
        "v_cndmask_b32_e32 reading VCC costs 16 cycles unless it directly follows the v_cmp that wrote VCC (pair: 4 + 4); one compare followed by four VCC selects "
        "costs 34-80 cycles against 18 for the e64 form on an SGPR pair. A select done as an exec-masked v_mov_b32 costs 2.4 cycles.",
        "Lanes switched off by EXEC cost the same issue cycles (no skipping of an empty half-wave).",
        f"The render kernels' own static opcode mixes cost {mix_mega:.2f} (k_megakernel) and {mix_trav:.2f} (k_wf_extend) cycles per instruction at 6 waves per SIMD: "
        "THAT is the denominator of valu_issue_busy, not an assumed 4 (round 1) and not the guide's 2.",
        "Counters: SQ_INSTS_VALU is the exact wave-instruction count; SQ_ACTIVE_INST_VALU counts 1 per 2- or 4-cycle instruction and 2 per 8-cycle one (it cannot tell "
        "the classes apart, so it is no busy measure); SQ_THREAD_CYCLES_VALU = active lanes x SQ_ACTIVE_INST_VALU units (64 per full instruction, 32 with half the "
        "wave masked off, 128 per v_rcp): lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); kernel cycles = GRBM_GUI_ACTIVE / 8.",
    ],
    "valu_ceiling_cycles_per_instruction": {"k_megakernel_static_mix": round(mix_mega, 3), "k_wf_extend_static_mix": round(mix_trav, 3),
                                            "waves_per_simd": 6, "mix_mega": plain.get("mix_mega", "")[:1200], "mix_trav": plain.get("mix_trav", "")[:1200]},
    "instruction_classes_at_6_waves_per_simd": classes,
    "cycles_per_instruction_per_simd": table,
    "counters_on_known_instruction_counts": counters,
    "shader_clock_ghz_under_load": round(ops["static VALU mix of k_megakernel"][6]["shader_clock_ghz"], 3),
}
print(json.dumps(out, indent=1))
