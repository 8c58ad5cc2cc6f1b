"""A counter of a rocprofv3 --pmc pass (p_counter_collection.csv) summed per kernel: launches, total, per launch.
usage: tools/diag/pmc_by_kernel.py gpurun_out/pmc_<tag>_<i>/p_counter_collection.csv"""
import csv,sys,collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: [0,0.0])
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][-46:]
    agg[(k,r["Counter_Name"])][0]+=1
    agg[(k,r["Counter_Name"])][1]+=float(r["Counter_Value"])
for (k,c),(n,v) in sorted(agg.items(), key=lambda x:-x[1][1])[:12]:
    print(f"{k:48s} {c:12s} launches {n:6d} total {v:.4g} per launch {v/n:.4g}")
