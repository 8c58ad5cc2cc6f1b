set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_EA0_RD[A-Z0-9_]*|TCC_[A-Z0-9_]*128[A-Z0-9_]*|TCC_EA[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/tcc_ea_names.txt || true
cat gpurun_out/tcc_ea_names.txt; echo
for mib in 512 2048 16384 65536; do timeout -k 10 120 tools/probes/hbm_gather $mib; done > gpurun_out/r02_hbm_gather_sizes.txt 2>&1
cat gpurun_out/r02_hbm_gather_sizes.txt
tools/run_pmc_tag.sh c2 "--steps 4 --warmup 1" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"
tools/run_pmc_tag.sh c5 "--workload c5 --steps 4 --warmup 1" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
