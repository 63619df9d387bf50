#!/bin/bash
# One run of the probes behind DESIGN.md's experiment paragraphs -> gpurun_out/r02/experiments.txt (copied to profiles/r02_experiments.txt)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02 && out=gpurun_out/r02/experiments.txt && : > $out
sec() { echo; echo "##### $1"; }
{
sec "refill threshold: workloads on which it matters (scripts/refill_matrix.sh), adaptive policy (default build)"; timeout -k 10 300 bash scripts/refill_matrix.sh
sec "wavefront schedules on 1/2, 1/4, 1/8 of the 1080p frame, 64 spp (scripts/tile_modes.sh)"; timeout -k 10 400 bash scripts/tile_modes.sh
sec "wavefront: samples per launch through the dynamic queue, by frame size (scripts/requeue_probe.sh)"; timeout -k 10 400 bash scripts/requeue_probe.sh
sec "decoupled traversal / shading megakernel RT_MEGA_TS=1 (scripts/ts_probe.sh)"; timeout -k 10 200 bash scripts/ts_probe.sh
sec "postponed leaves RT_POSTPONE=1 (librt_exp_postpone.so as variant pp) against the default build: PMC (scripts/variant_pmc.sh) and rates (scripts/ab.sh)"; timeout -k 10 300 bash scripts/variant_pmc.sh main pp; timeout -k 10 200 bash scripts/ab.sh 16 both main pp
sec "cost-ordered second launch RT_WF_LPT: Cornell box 1024x1024 64 spp, config 5 tile"
for lpt in 1 0; do echo "RT_WF_LPT=$lpt cornell: $(RT_WF_LPT=$lpt SCENE=cornell RES=1024x1024 timeout -k 10 100 python3 scripts/scene_probe.py 64 | grep Wave | cut -c21-48)"; echo "RT_WF_LPT=$lpt config 5 tile: $(RT_WF_LPT=$lpt timeout -k 10 200 python3 scripts/big_frames.py | grep "Wave.*config" | cut -c56-)"; done
} >> $out 2>&1
tail -5 $out
