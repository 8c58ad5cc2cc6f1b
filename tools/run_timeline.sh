#!/bin/bash
# Per-wave timeline + lane-efficiency of each trace launch of a frame (diagnostic variant library). On the GPU box.
# usage: tools/run_timeline.sh variant_name [iterations...]
v=$1; shift
for i in "$@"; do
  LRL_TIMELINE=$PWD/gpurun_out/tl_$i.bin LRL_TIMELINE_ITER=$i LRLTRACE_LIB=$PWD/raytracing_amd/variants/$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --steps 2 --warmup 1 $BENCH_ARGS > gpurun_out/tl_$i.json 2> gpurun_out/tl_$i.err || { echo "iter $i FAILED"; tail -3 gpurun_out/tl_$i.err; exit 1; }
  echo "== iteration $i"; python tools/timeline.py gpurun_out/tl_$i.bin 10 6; python tools/swarm_stats.py gpurun_out/tl_$i.bin
done
