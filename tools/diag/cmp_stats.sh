#!/bin/bash
# on the GPU box: per-launch averages of the shade kernels on C3 and C5 for the library LRLTRACE_LIB names (A/B of builds: tools/build_variant.sh)
# usage: TAG=name [LRLTRACE_LIB=...] tools/diag/cmp_stats.sh
for wl in "c3 --steps 3" "c5 --steps 3"; do
  set -- $wl
  bash tools/run_stats.sh x_${TAG}_$1 --workload $wl > /dev/null
  echo "$TAG $1"; grep -E "wfShadeKernel<(true|false), false>" gpurun_out/x_${TAG}_$1.txt | cut -c1-60,100-170
  rm -rf gpurun_out/x_${TAG}_$1
done
