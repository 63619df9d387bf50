#!/bin/bash
# Round 4, item 1: does the dispatcher spread the chain regime's 510 workgroups evenly (2 per CU)? Unused dynamic LDS caps the
# workgroups a CU can hold: 28 KiB of padding on top of the kernel's 52 KiB leaves room for two.
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
for PAD in 0 28672 60000; do
  echo "== RT_MEGA_LDS_PAD=$PAD"
  RT_MEGA_LDS_PAD=$PAD python3 scripts/tile_scaling.py 64 mega 8,4,2
  RT_KERNEL_STATS=1 RT_MEGA_LDS_PAD=$PAD python3 scripts/tile_scaling.py 64 mega 8 2>&1 | grep -E "waves|timeline" | tail -2
done
