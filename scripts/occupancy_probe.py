"""Same megakernel binary at 2 / 4 / 6 resident waves per SIMD (unused dynamic LDS limits the workgroups per CU, the persistent
grid is sized to match): does throughput follow the number of waves (latency-bound) or saturate (issue-bound)?
   python3 scripts/occupancy_probe.py [SPP]      (GPU box; each configuration runs in its own process: the knobs are read at renderer creation)"""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
spp = sys.argv[1] if len(sys.argv) > 1 else "8"
for occ, pad in ((2, 40 * 1024), (4, 12 * 1024), (6, 0)):
    env = dict(os.environ, RT_MEGA_OCC=str(occ), RT_MEGA_LDS_PAD=str(pad), RT_KERNEL_STATS=os.environ.get("RT_KERNEL_STATS", "0"))
    print(f"--- {occ} waves per SIMD (LDS pad {pad} B)", flush=True)
    subprocess.run([sys.executable, str(REPO / "scripts" / "quick_bench.py"), spp, "mega"], env=env, check=False)
