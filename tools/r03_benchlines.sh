# the round's bench lines (on the GPU box): gpurun_out/r03_bench_<name>.json
set -e
run() { n=$1; shift; timeout -k 10 500 python bench.py "$@" > gpurun_out/r03_bench_$n.json 2> gpurun_out/r03_bench_$n.err || { echo "$n FAILED"; tail -n 3 gpurun_out/r03_bench_$n.err; exit 1; }
  python3 - $n <<'PY'
import json,sys
j=json.loads(open(f"gpurun_out/r03_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1]); r=j["roofline"]
print(f'{sys.argv[1]:22s} ms/step {j["ms_per_step"]:9.3f}  Mrays/s {j["value"]:9.1f}  trace ms {r["kernel_ms"]:8.3f}  bound {r["bound"]} frac {r["frac"]}  cpu {j.get("cpu_baseline",{}).get("value")}')
PY
}
run default
run c1 --workload c1
run c3 --workload c3
run c3_device --workload c3 --builder device --no-cpu-baseline
run c4_1gpu --workload c4 --no-cpu-baseline
run c5 --workload c5 --no-cpu-baseline
run c2_device --builder device --no-c5 --no-cpu-baseline
run c2_median --builder median --no-c5 --no-cpu-baseline
run c2_parity --mode parity --no-c5 --no-cpu-baseline
run c2_diffuse --diffuse --no-c5 --no-cpu-baseline
run grid9_mirror --grid 9 --builder device --no-c5 --no-cpu-baseline --steps 10
run c3_deterministic --workload c3 --tuning deterministic=1 --no-cpu-baseline
run grid9_callers_tree --grid 9 --builder reference --no-c5 --no-cpu-baseline --steps 10
