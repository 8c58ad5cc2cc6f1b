# (the PMC passes pin LrlTuning.phasedTrips to what the library's own trial picks for the workload: under the profiler its timing is too
# noisy to pick the same way in every pass)
# the round's final measurement pass (on the GPU box): kernel statistics, PMC passes, bench lines
set -e
tools/run_stats.sh r02_c2_stats --steps 20
tools/run_stats.sh r02_c3_stats --workload c3 --steps 3
tools/run_stats.sh r02_c5_stats --workload c5 --steps 3
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"
tools/run_pmc_tag.sh c2 "--steps 4 --warmup 10 --tuning phasedTrips=2" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$SQ1" "$SQ2"
tools/run_pmc_tag.sh c5 "--workload c5 --steps 4 --warmup 8 --tuning phasedTrips=2" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$SQ1" "$SQ2"
tools/run_pmc_tag.sh c3 "--workload c3 --steps 2 --warmup 8 --tuning phasedTrips=1" "$SQ1"
