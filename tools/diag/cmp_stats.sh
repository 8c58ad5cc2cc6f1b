#!/bin/bash
# usage: cmp_stats.sh tag  (LRLTRACE_LIB picks the library)
for wl in "c3 --steps 3" "c5 --steps 3"; do
  set -- $wl
  bash tools/run_stats.sh x_$TAG_$1 --workload $wl > /dev/null
  grep -E "wfShadeKernel<(true|false), false>|wfGenerateKernel<false>|wfAccumulate" gpurun_out/x_$TAG_$1.txt | cut -c1-60,100-170
  rm -rf gpurun_out/x_$TAG_$1
done
