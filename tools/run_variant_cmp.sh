#!/bin/bash
# A/B of kernel-variant libraries (raytracing_amd/variants/<name>.so) against the built one (on the GPU box).
# usage: TUNING=key=value,... tools/run_variant_cmp.sh name...
T=${TUNING:+--tuning $TUNING}
for wl in "c2 --steps 40 --warmup 5" "c3 --steps 3 --warmup 2" "c5 --steps 3 --warmup 1" "c4 --steps 2 --warmup 1"; do
  for lib in base "$@"; do
    if [ $lib = base ]; then unset LRLTRACE_LIB; else export LRLTRACE_LIB=$PWD/raytracing_amd/variants/$lib.so; fi
    timeout -k 10 240 python bench.py --no-cpu-baseline --no-c5 --workload $wl $T > gpurun_out/cmp_${lib}.json 2> gpurun_out/cmp_${lib}.err || { echo "$lib $wl FAILED"; tail -n 3 gpurun_out/cmp_${lib}.err; exit 1; }
    python3 - "$lib" "$wl" gpurun_out/cmp_${lib}.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f'{sys.argv[1]:12s} {sys.argv[2]:28s} ms/step {j["ms_per_step"]:9.3f}  Mrays/s {j["value"]:9.1f}  trace ms {j["roofline"]["kernel_ms"]:9.3f}')
PY
  done
done
