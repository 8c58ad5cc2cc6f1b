#!/bin/bash
# on the GPU box: selected tests, then bench lines; usage: tools/r03_quick.sh "<pytest -k expression>" [bench args per line...]
mkdir -p gpurun_out
k=$1; shift
if [ -n "$k" ]; then
  timeout -k 10 900 python3 -u -m pytest tests -m gpu -q -x --timeout 420 --timeout-method thread -p no:cacheprovider -k "$k" > gpurun_out/r3_quick_tests.log 2>&1
  rc=$?; tail -n 15 gpurun_out/r3_quick_tests.log; echo "pytest rc=$rc"
  if [ $rc -ne 0 ]; then exit $rc; fi
fi
i=0
for args in "$@"; do
  i=$((i+1))
  timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-c5 $args > gpurun_out/r3_quick_$i.json 2> gpurun_out/r3_quick_$i.err || { echo "bench $i failed"; tail -n 5 gpurun_out/r3_quick_$i.err; exit 1; }
  python3 - gpurun_out/r3_quick_$i.json "$args" <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=l["roofline"]
print(sys.argv[2], "| ms/step", l["ms_per_step"], "Mrays/s", l["value"], "kernel_ms", r["kernel_ms"], "upload_s", l["config"]["scene_upload_s"], "phased", l["config"]["schedule"]["phasedTrips"], "useful", (r.get("valu_issue") or {}).get("useful_lane_frac"))
PY
done
