#!/bin/bash
# One rocprofv3 --pmc pass per argument (a quoted, space-separated counter list) over a short bench.py run; on the GPU box.
# usage: tools/run_pmc.sh "SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" ...   (extra bench args via BENCH_ARGS)
# Each pass gets its own run and its own kill timer (a pass that asks for too many counters of a block hangs).
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for list in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc$i
  timeout -s KILL 150 rocprofv3 --pmc $list -d gpurun_out/pmc$i -o p --output-format csv -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 $BENCH_ARGS > gpurun_out/pmc$i.json 2> gpurun_out/pmc$i.err || { echo "pass $i failed"; tail -3 gpurun_out/pmc$i.err; exit 1; }
  echo "pass $i ok: $list"
done
