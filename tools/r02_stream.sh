set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane_refill" --timeout 300 --timeout-method thread 2>&1 | tail -8
tools/run_tunings.sh "--workload c3 --steps 3 --warmup 2" s3 "-" "streamWords=4" "streamWords=8" "streamWords=16" "streamWords=8,refillMin=8" "streamWords=8,refillMin=32"
tools/run_tunings.sh "--workload c5 --steps 3 --warmup 1" s5 "-" "streamWords=8" "streamWords=16" "streamWords=32,refillMin=24"
tools/run_tunings.sh "--steps 30 --warmup 5" s2 "-" "streamWords=2" "streamWords=4" "streamWords=8"
