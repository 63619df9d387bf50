cd $GRAFT_REPO_ROOT
for st in 1 2 3; do for occ in 2 3 4 6; do
  echo "streams $st finish_occ $occ: $(RT_WF_STREAMS=$st RT_WF_FINISH_OCC=$occ RT_WF_FINISH_DEPTH=0 python3 scripts/quick_bench.py 8 wave 2>&1 | grep Wavefront)"
done; done
