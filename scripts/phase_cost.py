"""Dynamic VALU instruction mix and issue ceiling of the traversal kernels from their three step kinds (scripts/phase_mix.hip compiled with the
library's flags, assembly only) weighted with the measured steps per ray (RT_KERNEL_STATS=1 log):
   python3 scripts/phase_cost.py [stats.log] -> JSON: per phase instructions / calibrated cycles per step, per ray totals, ceiling cycles per instruction"""
import collections, json, re, subprocess, sys, tempfile
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
FLAGS = "-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero " \
        "-mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-result -fno-slp-vectorize --cuda-device-only -S".split()
calib = json.loads((REPO / "profiles" / "r02_valu_calibration.json").read_text())["cycles_per_instruction_per_simd"]


def op_cost(op):
    """calibrated issue cycles of one wave64 instruction at 6 waves per SIMD; unmeasured opcodes take their class's cost"""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    for name, v in calib.items():
        if name.split()[0] == base and not name.startswith(("pair", "group", "static")) and "(" not in name.replace(base, "", 1)[:2]:
            return v["w6"]
    two = {"v_subrev_f32", "v_mac_f32", "v_not_b32", "v_subrev_u32"}
    eight = {"v_rcp_iflag_f32", "v_exp_f32", "v_log_f32"}
    if op == "v_cndmask_b32_e32":
        return 8.0  # VCC form: 4 behind its compare, 16 otherwise
    return 2.15 if base in two else 8.1 if base in eight else 4.15


with tempfile.TemporaryDirectory() as d:
    s_file = Path(d) / "phase.s"
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, str(REPO / "scripts" / "phase_mix.hip"), "-o", str(s_file)], check=True, capture_output=True)
    text = s_file.read_text().splitlines()
# per probe kernel: the loop body = the instructions from the loop's header label to the backward branch that closes it (the loop that
# contains the "step boundary" marker); the prologue (trav_begin, LDS fill) and the epilogue are left out
def kernel_lines(name):
    out, on = [], False
    for ln in text:
        if re.match(rf"^{name}:", ln):
            on = True
            continue
        if on:
            if "s_endpgm" in ln:
                break
            out.append(ln)
    return out


def loop_body(lines, marker):
    """all basic blocks of the loop that contains `marker`, by the compiler's own block annotations
    ('=>This Inner Loop Header' on the header label, 'in Loop: Header=BBn_m' on the others)"""
    blocks, cur = [], {"label": None, "note": "", "lines": []}
    for ln in lines:
        m = re.match(r"^\.L(BB\d+_\d+):(.*)$", ln)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "note": m.group(2), "lines": []}
        else:
            cur["lines"].append(ln)
    blocks.append(cur)
    owner = next(b for b in blocks if any(marker in l for l in b["lines"]))
    m = re.search(r"Header=(BB\d+_\d+)", owner["note"])
    header = m.group(1) if m else owner["label"]
    body = []
    for b in blocks:
        if b["label"] == header or f"Header={header} " in b["note"] + " ":
            body.extend(b["lines"])
    return body


out = {"source": "scripts/phase_cost.py: scripts/phase_mix.hip (trav_inner / trav_leaf<true> / one shading round as separate kernels, the library's compile flags, "
                 "assembly only; loop bodies only) priced with profiles/r02_valu_calibration.json; steps per ray from an RT_KERNEL_STATS=1 frame",
       "phases": {}}
for nme, marker in (("phase_inner", "phase_inner step boundary"), ("phase_leaf", "phase_leaf step boundary"), ("phase_shade", "phase_shade round boundary")):
    body = loop_body(kernel_lines(nme), marker)
    h = collections.Counter(t[0] for t in (ln.split() for ln in body) if t and t[0].startswith("v_"))
    salu = sum(1 for t in (ln.split() for ln in body) if t and t[0].startswith("s_") and not t[0].startswith("s_waitcnt"))
    n = sum(h.values())
    cyc = sum(op_cost(o) * c for o, c in h.items())
    out["phases"][nme] = {"valu_instructions": n, "salu_instructions": salu, "issue_cycles": round(cyc, 1), "cycles_per_instruction": round(cyc / max(n, 1), 3),
                          "top_opcodes": ", ".join(f"{o} {c}" for o, c in h.most_common(14))}
stats = None
if len(sys.argv) > 1:
    for ln in open(sys.argv[1]):
        m = re.search(r"rays (\d+) \| inner steps (\d+) \(([\d.]+) lanes/step\) \| leaf steps (\d+) \(([\d.]+) lanes/step\) \| shade/refill rounds (\d+) \(([\d.]+)", ln)
        if m and "waves" not in ln:
            stats = [float(x) for x in m.groups()]
if stats:
    rays, si, li, sl, ll, ss, ls = stats
    # a whole-leaf step runs as many triangle tests as the largest leaf among its lanes (1..4, mean leaf 2.0 triangles): ~3.5 of the probe's 4
    w = {"phase_inner": si / rays, "phase_leaf": 0.875 * sl / rays, "phase_shade": ss / rays}
    ins = sum(w[k] * out["phases"][k]["valu_instructions"] for k in w)
    cyc = sum(w[k] * out["phases"][k]["issue_cycles"] for k in w)
    out["per_ray"] = {"wave_steps_per_ray": {k: round(v, 4) for k, v in w.items()}, "lanes_per_step": {"inner": li, "leaf": ll, "shade": ls},
                      "valu_instructions_per_ray": round(ins, 1), "valu_issue_cycles_per_ray": round(cyc, 1),
                      "dynamic_ceiling_cycles_per_instruction": round(cyc / ins, 3),
                      "share_of_issue_cycles": {k: round(w[k] * out["phases"][k]["issue_cycles"] / cyc, 3) for k in w}}
print(json.dumps(out, indent=1))
