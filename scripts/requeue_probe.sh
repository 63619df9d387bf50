#!/bin/bash
# streaming schedule of the wavefront renderer (RT_WF_SAMPLES_PER_LAUNCH = C > 1: breadth first through the dynamic queue) against one
# launch per sample (C = 1), by frame size: atrium, depth 10, 16 spp
cd $GRAFT_REPO_ROOT
for res in 960x540 1920x540 1920x1080 2560x1440 3840x2160; do
  for c in 1 4 16; do
    echo "== $res C=$c: $(RES=$res RT_WF_SAMPLES_PER_LAUNCH=$c python3 scripts/quick_bench.py 16 wave 2>&1 | grep Wavefront)"
  done
done
