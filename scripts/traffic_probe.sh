#!/bin/bash
# scripts/traffic_probe.sh SPP name1 name2 ...: fabric traffic of the megakernel per variant (FETCH_SIZE / WRITE_SIZE / TCC hit rate, separate --pmc passes) + frame time
spp=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = main ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/sycl-ray-tracer_amd/csrc/build/librt_var_$v.so; fi
  echo "=== $v"
  python3 scripts/quick_bench.py $spp mega 2>&1 | grep -v "^\[rt stats\]"
  D=gpurun_out/r03/traffic/$v; rm -rf $D; mkdir -p $D
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $D/a -- python3 bench.py --pmc-child megakernel --spp $spp > $D/a.out 2> $D/a.err
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $D/b -- python3 bench.py --pmc-child megakernel --spp $spp > $D/b.out 2> $D/b.err
  python3 - <<PY
import csv, glob, collections, json
c = collections.defaultdict(float)
for f in glob.glob("$D/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_megakernel" in r["Kernel_Name"]: c[r["Counter_Name"]] += float(r["Counter_Value"])
rays = json.loads(open("$D/a.out").read().strip().splitlines()[-1])["megakernel"]["rays"]
print("   fetch %.1f B/ray, write %.1f B/ray, L2 hit rate %.4f" % (c["FETCH_SIZE"] * 1024 / rays, c["WRITE_SIZE"] * 1024 / rays, c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
PY
done
