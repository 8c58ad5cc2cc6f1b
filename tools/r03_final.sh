#!/bin/bash
# the round's final measurement passes (on the GPU box): kernel statistics, PMC passes; usage: tools/r03_final.sh c2|c5
# (the PMC passes pin LrlTuning.phasedTrips to what the library's own trial picks for the workload: under the profiler its timing is too
# noisy to pick the same way in every pass)
set -e
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum"
TCP="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
if [ "$1" = c2 ]; then
  tools/run_stats.sh r03_c2_stats --steps 20 > /dev/null
  tools/run_stats.sh r03_c3_stats --workload c3 --steps 3 > /dev/null
  tools/run_pmc_tag.sh c2 "--steps 4 --warmup 10 --tuning phasedTrips=2" "FETCH_SIZE" "WRITE_SIZE" "$TCC" "$TCP" "$SQ1" "$SQ2"
  tools/run_pmc_tag.sh c3 "--workload c3 --steps 2 --warmup 8 --tuning phasedTrips=1" "$SQ1"
else
  tools/run_stats.sh r03_c5_stats --workload c5 --steps 3 > /dev/null
  tools/run_pmc_tag.sh c5 "--workload c5 --steps 4 --warmup 8 --tuning phasedTrips=2" "FETCH_SIZE" "WRITE_SIZE" "$TCC" "$TCP" "$SQ1" "$SQ2"
fi
