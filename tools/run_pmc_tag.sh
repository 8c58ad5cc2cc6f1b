#!/bin/bash
# One rocprofv3 --pmc pass per counter list over a short bench.py run, outputs under gpurun_out/pmc_<tag>_<i>; on the GPU box.
# usage: tools/run_pmc_tag.sh tag "<bench args>" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ...
# Each pass is a run of its own with its own kill timer (a pass that asks for too many counters of a block hangs).
tag=$1; args=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for list in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${tag}_$i
  timeout -s KILL 280 rocprofv3 --pmc $list -d gpurun_out/pmc_${tag}_$i -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-c5 $args > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.err || { echo "pass $i failed"; tail -n 3 gpurun_out/pmc_${tag}_$i.err; exit 1; }
  echo "pass $i ok: $list"
done
