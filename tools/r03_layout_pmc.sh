#!/bin/bash
# on the GPU box: HBM-side counters of C5's trace kernels with the four-wide tree laid out (treeLayout=0) and as built (1)
for L in 0 1; do
  tools/run_pmc_tag.sh c5L$L "--workload c5 --steps 2 --warmup 2 --tuning phasedTrips=2,treeLayout=$L" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for L in (0, 1):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/pmc_c5L{L}_*/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wfTrace" in k and "true" not in k.split("<")[1].split(",")[2 if "Lean" in k or "Group" in k else 0]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("treeLayout", L, {k: f"{v:.4g}" for k, v in tot.items()}, dict(n))
PY
