#!/bin/bash
# What binds the CHAIN REGIME? rocprofv3 --pmc passes (counters only) of the megakernel on 1/8 of the 1080p / 64 spp
# frame (259,200 pixels < 393,216 lanes) and, for comparison, on the full frame.
#   usage (GPU box, repo root): bash scripts/tile_pmc.sh <tag> [renderer=mega] [worlds="8 1"]
set -e
TAG=${1:-tile}; R=${2:-mega}; WORLDS=${3:-"8 1"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05/pmc_$TAG
mkdir -p $O
for WORLD in $WORLDS; do
  i=0
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $O/w${WORLD}_p$i -- python3 scripts/tile_scaling.py 64 $R $WORLD 1 > $O/w${WORLD}_p$i.out 2> $O/w${WORLD}_p$i.err || { echo "pass $i (1/$WORLD) failed"; tail -3 $O/w${WORLD}_p$i.err; }
  done
done
python3 - $O $WORLDS <<'PY' | tee $O/summary.txt
"""Summarises the counter passes of scripts/r04/tile_pmc.sh: per kernel and tile size, VALU busy, lane utilisation, L2 hit rate."""
import collections
import csv
import glob
import sys

root, worlds = sys.argv[1], sys.argv[2:]
for w in worlds:
    s = collections.defaultdict(lambda: collections.defaultdict(float))
    n_launch = collections.defaultdict(set)
    for f in glob.glob(f"{root}/w{w}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void rt::", "")
            s[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n_launch[k].add(r["Dispatch_Id"])
    for k, c in sorted(s.items()):
        if c.get("SQ_INSTS_VALU", 0) < 1e8:
            continue
        n = c["SQ_INSTS_VALU"]
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
        a, a2 = c.get("SQ_ACTIVE_INST_VALU", 0), c.get("SQ_ACTIVE_INST_VALU2", 0)
        print(f"1/{w} of the frame | {k} | launches {len(n_launch[k])}")
        print("   VALU instr %.4g | kernel cycles/instr/SIMD %.3f | VALU busy 4(A-A2)/(1024 cyc) = %.3f | paired share %.3f | lane utilisation %.3f"
              % (n, cyc * 1024 / n if n else 0, 4 * (a - a2) / 1024 / cyc if cyc else 0, 2 * a2 / n if n else 0,
                 c.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * a) if a else 0))
        hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        print("   L2 hit %.3f | fetch %.3g KiB write %.3g KiB | waves %.4g | SALU/VALU %.3f | LDS instr/VALU %.3f | VMEM_RD/VALU %.4f"
              % (hit / (hit + miss) if hit + miss else 0, c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0), c.get("SQ_WAVES", 0),
                 c.get("SQ_INSTS_SALU", 0) / n, c.get("SQ_INSTS_LDS", 0) / n, c.get("SQ_INSTS_VMEM_RD", 0) / n))
        # the vector-memory path (the TA_* busy counters abort rocprofv3 on this image): mean latency of an L1 access and of an L1 -> L2 read,
        # share of the kernel's cycles the L1s stall on pending misses, LDS bank conflicts, vector-memory instructions in flight per wave
        print("   vector memory: L1 latency %.0f cycles per L2 read issued, L2 read latency %.0f cycles | L1 stalled on pending misses %.3f of the kernel's cycles (per CU) | "
              "LDS bank-conflict cycles / active cycles %.3f | vector-memory instructions in flight per wave %.2f"
              % (c.get("TCP_TCP_LATENCY_sum", 0) / max(c.get("TCP_TCC_READ_REQ_sum", 0), 1), c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(c.get("TCP_TCC_READ_REQ_sum", 0), 1),
                 c.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / cyc if cyc else 0, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1),
                 c.get("SQ_INST_LEVEL_VMEM", 0) / max(c.get("SQ_WAVE_CYCLES", 0), 1)))
        # the vector L1 looks one cache line up per cycle and CU: a lane's 16-byte load is one access, four of them fetch a node
        l1 = c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)
        if l1:
            rays = {"8": 135898371, "4": 272335626, "2": 545124372, "1": 1082550484}.get(w, 0) * len(n_launch[k])  # (the bench frame's tiles, megakernel seeds)
            print("   vector L1: %.3f tag look-ups per cycle and CU (1 = its rate), %.1f per ray | read tag-conflict stalls %.3f, address stalls towards the TA %.3f of the cycles"
                  % (l1 / 256 / cyc if cyc else 0, l1 / rays if rays else 0, c.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0) / 256 / cyc if cyc else 0,
                     c.get("TCP_TCP_TA_ADDR_STALL_CYCLES_sum", 0) / 256 / cyc if cyc else 0))
        wc = c.get("SQ_WAVE_CYCLES", 0)
        print("   per wave-cycle: waiting (any) %.3f, issuing (any) %.3f | SQ_BUSY_CYCLES %.4g | LDS wait %.3g active %.3g | VMEM active %.3g"
              % (c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0, c.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else 0, c.get("SQ_BUSY_CYCLES", 0),
                 c.get("SQ_WAIT_INST_LDS", 0), c.get("SQ_ACTIVE_INST_LDS", 0), c.get("SQ_ACTIVE_INST_VMEM", 0)))
PY
