#!/usr/bin/env python3
"""Per-dispatch durations of the LAST frame in a rocprofv3 --kernel-trace CSV (tools/run_stats.sh), in launch order.
usage: tools/frame_trace.py gpurun_out/<name> [dispatches-per-frame-guess]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: from the last wfGenerateKernel<false> on
last = max(i for i, r in enumerate(rows) if "wfGenerateKernel<false>" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    name = r["Kernel_Name"].split("(")[0].replace("void lrl_fast::", "")
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  {name}')
