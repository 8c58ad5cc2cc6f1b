#!/usr/bin/env python3
"""Folds the rocprofv3 outputs of tools/run_stats.sh + tools/run_pmc.sh (under gpurun_out/) into the committed
summaries profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and profiles/r01_traffic.json (read by bench.py).
usage: tools/summarize_profile.py tag stats_dir "note" [frames_in_pmc_run]
Counters are summed over the launches of one frame per kernel (a frame = one pass of the pipeline = several launches).
HBM traffic per frame follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies the 128-B requests of 16-B-per-lane loads at 64 B, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, shutil, sys
tag, stats_dir, note = sys.argv[1], sys.argv[2], sys.argv[3]
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 10         # run_pmc.sh: --steps 4 --warmup 1, and bench.py renders 5 more for its per-kernel timing
shutil.copyfile(glob.glob(f"gpurun_out/{stats_dir}/*kernel_stats.csv")[0], f"profiles/{tag}_kernel_stats.csv")
per = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for f in sorted(glob.glob("gpurun_out/pmc*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wf" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        per[short][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[short] = {x: r[x] for x in ("Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
out = {"note": note, "frames_profiled": frames, "per_frame": {}}
trace = collections.defaultdict(float)
for k, c in per.items():
    out["per_frame"][k] = {"dispatch": meta[k], "counters": {n: v / frames for n, v in sorted(c.items())}}
    if "Trace" in k and "<true>" not in k:
        for n, v in c.items():
            trace[n] += v / frames
d = {}
if "SQ_INSTS_VALU" in trace:
    d["instructions_per_frame"] = {n: trace[n] for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS") if n in trace}
if "SQ_THREAD_CYCLES_VALU" in trace:
    d["valu_lane_utilisation"] = trace["SQ_THREAD_CYCLES_VALU"] / (trace["SQ_ACTIVE_INST_VALU"] * 64)
    d["wave_cycles_split"] = {n: trace[n] / trace["SQ_WAIT_ANY"] * 0 + trace[n] for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
if "TCC_HIT_sum" in trace:
    d["l2_hit_rate"] = trace["TCC_HIT_sum"] / max(1.0, trace["TCC_HIT_sum"] + trace["TCC_MISS_sum"])
    d["l1_hit_rate"] = 1 - trace["TCP_TCC_READ_REQ_sum"] / max(1.0, trace["TCP_TOTAL_CACHE_ACCESSES_sum"])
if "FETCH_SIZE" in trace:
    d["hbm_bytes_per_frame"] = int(trace["FETCH_SIZE"] * 1024 * 2 + trace.get("WRITE_SIZE", 0) * 1024)
    d["hbm_read_bytes_per_frame"] = int(trace["FETCH_SIZE"] * 1024 * 2)
    d["hbm_write_bytes_per_frame"] = int(trace.get("WRITE_SIZE", 0) * 1024)
out["trace_kernels_per_frame"] = d
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(d, indent=1))
if "hbm_bytes_per_frame" in d and len(sys.argv) > 5:
    wl = json.loads(sys.argv[5])
    wl["hbm_bytes_per_frame"] = d["hbm_bytes_per_frame"]
    wl["source"] = f"profiles/{tag}_pmc.json"
    json.dump(wl, open("profiles/r01_traffic.json", "w"), indent=1)
