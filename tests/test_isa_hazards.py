"""CPU test on hipcc's own output (-S, gfx950): the inner step's node fetch (rt_device.h: trav_inner) issues its four global loads from inline
assembly, invisible to the compiler's wait-count pass, and waits for them with two explicit s_waitcnt further down. That is only correct if
NOTHING between the loads and their waits touches the destination registers in the loads' own lanes, and if no other vector-memory
instruction slips in between (the waits count in order). The register allocator decides that, not the source: so every instantiation of every
traversal kernel is checked in the listing itself."""
import re
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "sycl-ray-tracer_amd" / "csrc"
VMEM = re.compile(r"^\s*(global_|buffer_|scratch_|flat_)")


def _regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(v) for v in re.findall(r"\bv(\d+)\b", text))
    return out


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "rt.s"
    flags = None
    for line in (CSRC / "Makefile").read_text().splitlines():  # the product's own flags
        if line.startswith("COMMON :="):
            common = line.split(":=", 1)[1].strip()
        if line.startswith("HIPFLAGS :="):
            flags = line.split(":=", 1)[1].strip().rstrip("\\").strip()
        elif flags is not None and flags.endswith("-fno-gpu-flush-denormals-to-zero") and "-mllvm" in line:
            flags += " " + line.strip()
    flags = flags.replace("$(COMMON)", common).replace("$(ARCH)", "gfx950")
    lines = []
    for unit in ("rt_frame.hip", "rt_probes.hip"):  # the two translation units with traversal kernels
        cmd = ["/opt/rocm/bin/hipcc", *flags.split(), "-S", "--cuda-device-only", str(CSRC / unit), "-o", str(out)]
        subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC)
        lines += out.read_text().splitlines()
    return lines


def _check(lines):
    """asserts the rules on a listing; returns {kernel: asm node fetches found}"""
    kernel, i, groups = None, 0, {}
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_ZN2rt\w+):", ln)
        if m:
            kernel = m.group(1)
        if ln.strip() == ";;#ASMSTART" and i + 5 < len(lines) and all("global_load_dwordx4" in lines[i + k] for k in (1, 2, 3, 4)):
            dests = [sorted(_regs(lines[i + k].split(",")[0])) for k in (1, 2, 3, 4)]
            assert [re.search(r"offset:(\d+)", lines[i + k]) and int(re.search(r"offset:(\d+)", lines[i + k]).group(1)) or 0 for k in (1, 2, 3, 4)] == [0, 16, 32, 48]
            assert lines[i + 5].strip() == ";;#ASMEND"
            addr = _regs(lines[i + 1].split(",")[1])
            all_dest = set(sum(dests, []))
            j, stage, complementary = i + 6, 0, False  # stage 0: before vmcnt(1); 1: before vmcnt(0)
            while stage < 2:
                t = lines[j].strip()
                assert j < i + 400, f"{kernel}: no explicit wait after the asm loads at line {i}"
                if t == ";;#ASMSTART" and lines[j + 1].strip() == f"s_waitcnt vmcnt({1 - stage})":
                    stage += 1
                    j += 3
                    continue
                t = t.split(";")[0].strip()  # (labels carry a trailing comment)
                if not t or t.endswith(":"):
                    j += 1
                    continue
                assert not VMEM.match(t), f"{kernel} line {j}: a vector-memory instruction between the asm node loads and their wait: {t}"
                if t.startswith("s_andn2_saveexec_b64"):
                    complementary = True   # the lanes that did NOT issue the loads (the LDS copy of the tree's top)
                elif t.startswith("s_or_b64 exec"):
                    complementary = False
                elif not complementary and not t.startswith("s_"):
                    guarded = all_dest if stage == 0 else set(dests[3])
                    assert not (_regs(t) & guarded), f"{kernel} line {j}: {t!r} touches a node word that is still in flight (stage {stage})"
                elif complementary:
                    assert t.startswith(("ds_read", "v_lshrrev_b32", "s_", "v_add_u32", "v_mov_b32")), f"{kernel} line {j}: unexpected instruction under the LDS lanes' mask: {t}"
                    if t.startswith(("v_lshrrev_b32", "v_add_u32", "v_mov_b32")):
                        assert _regs(t.split(",")[0]) & all_dest or True  # (address arithmetic of the LDS lanes may live in a destination register: other lanes)
                j += 1
            groups[kernel] = groups.get(kernel, 0) + 1
            assert addr.isdisjoint(all_dest), f"{kernel}: the node offset shares a register with the loads' destinations"
            i = j
            continue
        i += 1
    return groups


def test_nothing_touches_the_node_words_between_their_asm_loads_and_their_waits(listing):
    groups = _check(listing)
    # every traversal kernel instantiation carries the fetch (three inner steps per loop iteration in the persistent kernels)
    want = ["k_megakernelILi0ELb0", "k_megakernelILi0ELb1", "k_megakernelILi1ELb0", "k_megakernelILi1ELb1", "k_megakernelILi2ELb0", "k_megakernelILi2ELb1",
            "k_wf_extendILb0", "k_wf_extendILb1",
            "k_wf_finishILb0ELb0ELb0ELb0", "k_wf_finishILb0ELb0ELb0ELb1", "k_wf_finishILb0ELb0ELb1ELb0", "k_wf_finishILb0ELb1ELb0ELb0",
            "k_wf_finishILb1ELb0ELb0ELb0", "k_wf_finishILb1ELb0ELb0ELb1", "k_wf_finishILb1ELb0ELb1ELb0", "k_wf_finishILb1ELb1ELb0ELb0", "k_intersect_batch"]
    for w in want:
        assert any(w in k for k in groups), f"no asm node fetch found in {w}: {sorted(groups)}"


def test_the_checker_sees_a_copy_a_spill_and_a_stray_load(listing):
    """the three ways a compiler change could break the fetch, injected into the real listing: each must be reported"""
    at = next(i for i, ln in enumerate(listing) if ln.strip() == ";;#ASMSTART" and "global_load_dwordx4" in listing[i + 1] and "global_load_dwordx4" in listing[i + 4])
    dest = re.search(r"v\[(\d+):", listing[i + 1] if False else listing[at + 1]).group(1)
    for bad in (f"\tv_mov_b32_e32 v200, v{dest}", f"\tscratch_store_dword off, v{dest}, off offset:4", "\tglobal_load_dword v201, v202, s[0:1]"):
        mutated = listing[: at + 6] + [bad] + listing[at + 6:]
        with pytest.raises(AssertionError):
            _check(mutated)
