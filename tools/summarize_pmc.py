#!/usr/bin/env python3
"""Collects the per-launch averages of the trace kernel from rocprofv3 --pmc passes (gpurun_out/pmc*/p_counter_collection.csv)
into one JSON under profiles/.  usage: tools/summarize_pmc.py out.json "note" [kernel-substring]"""
import collections, csv, glob, json, sys
out, note = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "renderKernel<false, false>"
c = {}
for f in sorted(glob.glob("gpurun_out/pmc*/p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
    c.update({k: sum(v) / len(v) for k, v in agg.items()})
d = {"note": note, "kernel": pat, "dispatch": meta, "counters_per_launch": c}
if "SQ_THREAD_CYCLES_VALU" in c:
    d["derived"] = {"valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64),
                    "wait_any_frac_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                    "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(1, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)),
                    "l1_hit_rate": 1 - c.get("TCP_TCC_READ_REQ_sum", 0) / max(1, c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1)),
                    "hbm_traffic_bytes_fetch_plus_write_KBx1024": (c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d["derived"], indent=1)); print(c)
