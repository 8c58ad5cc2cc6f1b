#!/bin/bash
# Runs bench.py (no CPU baseline) once per kernel-variant library under raytracing_amd/variants/ on the GPU box.
# usage (on the box): tools/run_variants.sh [bench args...]
mkdir -p gpurun_out/variants
for so in raytracing_amd/variants/*.so; do
  n=$(basename $so .so)
  LRLTRACE_LIB=$PWD/$so timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" > gpurun_out/variants/$n.json 2> gpurun_out/variants/$n.err || { echo "$n FAILED"; exit 1; }
  python - "$n" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/variants/{sys.argv[1]}.json"))
print(sys.argv[1], d["value"], "Mrays/s kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
PY
done
