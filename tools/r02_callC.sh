set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/pmc_probe_s
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum -d gpurun_out/pmc_probe_s -o p --output-format csv -- tools/probes/hbm_gather 4096 > gpurun_out/pmc_probe_s.txt 2>&1
PASS="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
rm -rf gpurun_out/pmc_c2_7 gpurun_out/pmc_c5_6
timeout -s KILL 280 rocprofv3 --pmc $PASS -d gpurun_out/pmc_c2_7 -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-c5 --steps 4 --warmup 1 > gpurun_out/pmc_c2_7.json 2> gpurun_out/pmc_c2_7.err
timeout -s KILL 280 rocprofv3 --pmc $PASS -d gpurun_out/pmc_c5_6 -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-c5 --workload c5 --steps 4 --warmup 1 > gpurun_out/pmc_c5_6.json 2> gpurun_out/pmc_c5_6.err
echo ok
