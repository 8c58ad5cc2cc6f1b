# the round's last bench lines (on the GPU box), after the fitted tree, the level-by-level build and the idle hand-over launches
set -e
run() { n=$1; shift; timeout -k 10 500 python bench.py "$@" > gpurun_out/r02_bench_$n.json 2> gpurun_out/r02_bench_$n.err || { echo "$n FAILED"; tail -n 3 gpurun_out/r02_bench_$n.err; exit 1; }
  python3 - $n <<'PY'
import json,sys
j=json.loads(open(f"gpurun_out/r02_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1]); r=j["roofline"]
print(f'{sys.argv[1]:22s} ms/step {j["ms_per_step"]:9.3f}  Mrays/s {j["value"]:9.1f}  trace ms {r["kernel_ms"]:8.3f}  bound {r["bound"]} frac {r["frac"]:.3f}  cpu {j.get("cpu_baseline",{}).get("value")}')
PY
}
run default
run c1 --workload c1
run c3 --workload c3
run c4_1gpu --workload c4 --no-cpu-baseline
run c5 --workload c5 --no-cpu-baseline
run c2_device --builder device --no-c5 --no-cpu-baseline
run c2_sah --builder sah --no-c5 --no-cpu-baseline
run c3_device --workload c3 --builder device --no-cpu-baseline
run c3_device_lbvh --workload c3 --builder device --tuning deviceBuilder=1 --no-cpu-baseline
