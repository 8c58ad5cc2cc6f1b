#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench.py run (on the GPU box); output under gpurun_out/<name>/
# usage: tools/run_stats.sh name [bench args...]
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/$name
timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d gpurun_out/$name -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-c5 "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err || { echo "stats run failed"; tail -n 5 gpurun_out/$name.err; exit 1; }
f=$(find gpurun_out/$name -name "*kernel_stats.csv" | head -n 1)
python3 - "$f" > gpurun_out/$name.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} {float(r["Percentage"]):6.2f}%')
PY
cat gpurun_out/$name.txt
