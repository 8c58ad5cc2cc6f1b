#!/bin/bash
# bench.py once per environment setting (on the GPU box). usage: tools/run_env.sh "A=1 B=2" "A=3" ...
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/env_$i.json 2> gpurun_out/env_$i.err || { echo "$e FAILED"; tail -3 gpurun_out/env_$i.err; exit 1; }
  python - "$e" gpurun_out/env_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print(sys.argv[1], "->", d["value"], "Mrays/s  ms/frame", d["ms_per_step"])
PY
done
