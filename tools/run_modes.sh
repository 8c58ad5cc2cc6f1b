#!/bin/bash
# bench.py once per trace-kernel mode (on the GPU box). usage: tools/run_modes.sh "mode[:hybrid_iter]" ...
for m in "$@"; do
  mode=${m%%:*}; it=${m#*:}; [ "$it" = "$m" ] && it=2
  LRL_TRACE=$mode LRL_HYBRID_ITER=$it timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/mode_$mode$it.json 2> gpurun_out/mode_$mode$it.err || { echo "$m FAILED"; tail -3 gpurun_out/mode_$mode$it.err; exit 1; }
  python - "$m" gpurun_out/mode_$mode$it.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], "Mrays/s  ms/frame", d["ms_per_step"])
PY
done
