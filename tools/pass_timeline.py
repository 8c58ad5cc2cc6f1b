#!/usr/bin/env python3
"""Per-launch timeline of the last pass of the last frame in a rocprofv3 --kernel-trace csv, and the last frame's time by kernel.
usage: tools/pass_timeline.py gpurun_out/<name>/s_kernel_trace.csv [passes per frame]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 11
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gen = [i for i, r in enumerate(rows) if "wfGenerateKernel<false>" in r["Kernel_Name"]]
last = rows[gen[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    short = r["Kernel_Name"].split("(")[0].replace("void lrl_fast::", "")[:56]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:9.1f} {short} grid={r.get('Grid_Size_X', r.get('Grid_Size'))}")
frame = rows[gen[-passes]:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in frame) / 1e6
span = (int(frame[-1]["End_Timestamp"]) - int(frame[0]["Start_Timestamp"])) / 1e6
print(f"last frame: kernels {busy:.2f} ms, span {span:.2f} ms")
agg = collections.defaultdict(float)
for r in frame:
    agg[r["Kernel_Name"].split("(")[0][-60:]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in sorted(agg.items(), key=lambda x: -x[1]):
    print(f"{v:8.2f} {k}")
