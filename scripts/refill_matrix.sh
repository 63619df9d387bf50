#!/bin/bash
# the workloads on which the refill threshold matters, both renderers (default library or RT_MI355X_LIB)
cd $GRAFT_REPO_ROOT
for spp in 64 16 4; do echo "== atrium 1080p $spp spp: $(timeout -k 10 100 python3 scripts/quick_bench.py $spp both | grep Renderer | cut -c1-12,62- | tr "\n" " ")"; done
echo "== atrium 8K 4 spp: $(RES=7680x4320 timeout -k 10 100 python3 scripts/quick_bench.py 4 both | grep Renderer | cut -c1-12,62- | tr "\n" " ")"
echo "== cornell 4K 16 spp: $(SCENE=cornell RES=3840x2160 timeout -k 10 100 python3 scripts/scene_probe.py 16 | cut -c1-12,34-48 | tr "\n" " ")"
echo "== cornell 1024 64 spp: $(SCENE=cornell RES=1024x1024 timeout -k 10 100 python3 scripts/scene_probe.py 64 | cut -c1-12,34-48 | tr "\n" " ")"
echo "== cube 4K 16 spp: $(SCENE=cube RES=3840x2160 timeout -k 10 100 python3 scripts/scene_probe.py 16 | cut -c1-12,34-48 | tr "\n" " ")"
echo "== cube 512 16 spp: $(SCENE=cube RES=512x512 timeout -k 10 100 python3 scripts/scene_probe.py 16 | cut -c1-12,34-48 | tr "\n" " ")"
