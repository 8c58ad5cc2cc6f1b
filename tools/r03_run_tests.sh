#!/bin/bash
# on the GPU box: the whole GPU suite (every failure listed, not only the first), then the default bench line when the suite ended normally
mkdir -p gpurun_out
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -q --timeout 420 --timeout-method thread -p no:cacheprovider "$@" > gpurun_out/r3_tests.log 2>&1
rc=$?
tail -n 25 gpurun_out/r3_tests.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 600 python3 bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
brc=$?
echo "bench rc=$brc"; tail -c 3000 gpurun_out/r3_bench_default.json; tail -n 5 gpurun_out/r3_bench_default.err
exit $brc
