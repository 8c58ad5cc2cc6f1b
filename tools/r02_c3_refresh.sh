# C3 again after the four-wide tree was fitted to the triangles' own boxes (on the GPU box): kernel statistics, the SQ pass, bench lines
set -e
tools/run_stats.sh r02_c3_stats --workload c3 --steps 3
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline > gpurun_out/c3_probe.json 2> gpurun_out/c3_probe.err
PH=$(python3 -c "import json;print(2 if json.loads(open('gpurun_out/c3_probe.json').read().strip().splitlines()[-1])['config']['schedule']['phasedTrips']==0 else 1)")
echo "triangle trips as the library's trial decides for C3 now: phasedTrips=$PH"
tools/run_pmc_tag.sh c3 "--workload c3 --steps 2 --warmup 8 --tuning phasedTrips=$PH" "$SQ1"
run() { n=$1; shift; timeout -k 10 500 python bench.py "$@" > gpurun_out/r02_bench_$n.json 2> gpurun_out/r02_bench_$n.err || { echo "$n FAILED"; tail -n 3 gpurun_out/r02_bench_$n.err; exit 1; }
  python3 - $n <<'PY'
import json,sys
j=json.loads(open(f"gpurun_out/r02_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1]); r=j["roofline"]
print(f'{sys.argv[1]:22s} ms/step {j["ms_per_step"]:9.3f}  Mrays/s {j["value"]:9.1f}  trace ms {r["kernel_ms"]:8.3f}  bound {r["bound"]} frac {r["frac"]:.3f}  cpu {j.get("cpu_baseline",{}).get("value")}')
PY
}
run c1 --workload c1
run c3 --workload c3
run c3_device --workload c3 --builder device --no-cpu-baseline
run default
