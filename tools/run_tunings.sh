#!/bin/bash
# bench.py once per --tuning setting (on the GPU box).  usage: tools/run_tunings.sh "<bench args>" tag "tuning1" "tuning2" ...   ("-" = defaults)
args=$1; tag=$2; shift 2
i=0
for t in "$@"; do
  i=$((i+1))
  if [ "$t" = "-" ]; then targ=""; else targ="--tuning $t"; fi
  timeout -k 10 240 python bench.py --no-cpu-baseline --no-c5 $args $targ > gpurun_out/tune_${tag}_$i.json 2> gpurun_out/tune_${tag}_$i.err || { echo "$t FAILED"; tail -n 3 gpurun_out/tune_${tag}_$i.err; exit 1; }
  python3 - "$t" gpurun_out/tune_${tag}_$i.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=j["roofline"]
print(f'{sys.argv[1]:40s} ms/step {j["ms_per_step"]:9.3f}  Mrays/s {j["value"]:9.1f}  trace ms {r["kernel_ms"]:9.3f}  launches {r["launches_per_frame"]:5d}  thr {j["config"]["schedule"].get("handoverTrips", [])[:5]} lanes {j["config"]["schedule"]["groupLanes"]} slots {j["config"]["schedule"]["contSlots"]} ph {j["config"]["schedule"].get("phasedTrips")}')
PY
done
