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
        wc = c.get("SQ_WAVE_CYCLES", 0)
        print("   per wave-cycle: waiting (any) %.3f, issuing (any) %.3f | SQ_BUSY_CYCLES %.4g | LDS wait %.3g active %.3g | VMEM active %.3g"
              % (c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0, c.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else 0, c.get("SQ_BUSY_CYCLES", 0),
                 c.get("SQ_WAIT_INST_LDS", 0), c.get("SQ_ACTIVE_INST_LDS", 0), c.get("SQ_ACTIVE_INST_VMEM", 0)))
