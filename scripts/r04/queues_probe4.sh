#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # $1 = comma list of renderers, rest env
  echo "== $*"
  L=$1; shift
  env "$@" python3 -c "
import sys, json, io, contextlib
sys.argv=['bench.py','--steps','3','--warmup','1','--no-live-pmc','--no-cpu-baseline','--no-pipelined']
import bench
bench.RENDERERS=tuple('$L'.split(','))
buf=io.StringIO()
with contextlib.redirect_stdout(buf): bench.main()
d=json.loads([l for l in buf.getvalue().splitlines() if l.startswith('{')][0]); print({k:(v['value'],v['ms_per_step']) for k,v in d['renderers'].items()})" 2>/dev/null
}
run wavefront_per_bounce,wavefront,megakernel RT_BENCH_FORCE_DIST=1
run wavefront,wavefront_per_bounce RT_BENCH_FORCE_DIST=1
run megakernel,wavefront_per_bounce RT_BENCH_FORCE_DIST=1
run wavefront_per_bounce_fused,wavefront_per_bounce RT_BENCH_FORCE_DIST=1
run wavefront_per_bounce,wavefront_per_bounce RT_BENCH_FORCE_DIST=1
