set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 120 tools/probes/hbm_gather > gpurun_out/r02_hbm_gather.txt 2>&1
cat gpurun_out/r02_hbm_gather.txt
rm -rf gpurun_out/pmc_probe_f gpurun_out/pmc_probe_w
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_probe_f -o p --output-format csv -- tools/probes/hbm_gather > gpurun_out/pmc_probe_f.txt 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d gpurun_out/pmc_probe_w -o p --output-format csv -- tools/probes/hbm_gather > gpurun_out/pmc_probe_w.txt 2>&1 || echo "rdreq pass failed"
echo probe done
timeout -k 10 500 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
tail -c 3000 gpurun_out/r02_bench.json
tools/run_stats.sh r02_c2_stats --steps 20 --warmup 3
tools/run_stats.sh r02_c3_stats --workload c3 --steps 3 --warmup 1
tools/run_stats.sh r02_c5_stats --workload c5 --steps 3 --warmup 1
