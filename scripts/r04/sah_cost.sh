#!/bin/bash
# Round 4, item 3b: the SAH leaf criterion's traversal-cost weight priced with the MEASURED step costs (an inner step ~118 vector
# instructions, a triangle test ~61 after the fused R5: ratio ~1.9) against the builder's generic 1.0: inner steps and triangle tests per
# ray (RT_KERNEL_STATS) and the frame time, atrium 1080p, 16 spp.
cd $GRAFT_REPO_ROOT
for C in 1.0 1.5 1.9 2.5; do
  echo "== RT_SAH_TRAV_COST=$C"
  RT_SAH_TRAV_COST=$C RT_KERNEL_STATS=1 python3 scripts/quick_bench.py 16 mega 2>&1 | grep -E "per ray|Mrays" | tail -2 | cut -c1-300
  RT_SAH_TRAV_COST=$C python3 scripts/quick_bench.py 16 mega 2>&1 | grep Mrays
done
