#!/bin/bash
# bench.py once per environment setting with fixed extra bench arguments (on the GPU box).
# usage: BENCH_ARGS="--workload c3 --steps 3 --warmup 1" tools/run_env_args.sh "A=1 B=2" "A=3" ...
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/enva_$i.json 2> gpurun_out/enva_$i.err || { echo "$e FAILED"; tail -3 gpurun_out/enva_$i.err; exit 1; }
  python - "$e" gpurun_out/enva_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print(sys.argv[1], "->", d["value"], "Mrays/s  ms/frame", d["ms_per_step"], "trace ms", d["roofline"]["kernel_ms"])
PY
done
