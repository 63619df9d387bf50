#!/bin/bash
# the wavefront renderer on 1/2, 1/4, 1/8 of the 1080p frame (64 spp): schedule by samples per launch / hand-over form / stream lanes
cd $GRAFT_REPO_ROOT
run() { echo "--- $1"; shift; env "$@" python3 scripts/tile_scaling.py 64 wave 2,4,8 | cut -c1-58,95-; }
run "default" X=1
run "per sample" RT_WF_SAMPLES_PER_LAUNCH=1
run "requeue 16" RT_WF_SAMPLES_PER_LAUNCH=16
run "requeue 64" RT_WF_SAMPLES_PER_LAUNCH=64
run "in-lane 64" RT_WF_SAMPLES_PER_LAUNCH=64 RT_WF_REQUEUE=0
run "in-lane 64, one stream" RT_WF_SAMPLES_PER_LAUNCH=64 RT_WF_REQUEUE=0 RT_WF_STREAMS=1
run "requeue 64, one stream" RT_WF_SAMPLES_PER_LAUNCH=64 RT_WF_STREAMS=1
echo "--- megakernel"; python3 scripts/tile_scaling.py 64 mega 2,4,8 | cut -c1-58,95-
