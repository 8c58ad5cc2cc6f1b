#!/usr/bin/env python3
"""Folds the rocprofv3 outputs of tools/run_stats.sh + tools/run_pmc_tag.sh (under gpurun_out/) into the committed summaries
profiles/<name>_kernel_stats.csv, profiles/<name>_pmc.json and the traffic file bench.py reads (profiles/<round>_traffic*.json).
The traffic file carries the sha256 of the kernel sources the passes were measured on (bench.csrc_hash(): run this script on the tree
the profiled library was built from) and the LrlTuning the passes pinned; bench.py withholds the roofline fraction when they are
not the sources of the build it times.
usage: tools/summarize_round.py name stats_dir pmc_tag frames traffic_file '<json: the keys bench.py matches>' "note" ["tuning of the passes"]
Counters are summed over the launches of one frame per kernel (frames = frames the profiled run rendered with the timed kernels:
warm-up + steps + the per-launch timing pass of bench.py).  HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md:
FETCH_SIZE and WRITE_SIZE are in KiB and taken in passes of their own; FETCH_SIZE counts a 128-B request as 64 B on gfx950.  The
factor this access shape needs (every lane its own 64-B record) is calibrated by tools/probes/hbm_gather.hip under the same counter
(profiles/r02_hbm_gather_probe.txt) and passed in as FETCH_FACTOR (environment, default 2 as the guide prescribes)."""
import collections, csv, glob, json, os, shutil, sys
name, stats_dir, pmc_tag, frames, traffic_file, match, note = sys.argv[1:8]
tuning = sys.argv[8] if len(sys.argv) > 8 else ""
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
frames = int(frames)
factor = float(os.environ.get("FETCH_FACTOR", "2"))
if stats_dir != "-":
    shutil.copyfile(glob.glob(f"gpurun_out/{stats_dir}/**/*kernel_stats.csv", recursive=True)[0], f"profiles/{name}_kernel_stats.csv")
def is_counting(kernel):
    """The CNT template argument of a wf kernel: wfTraceKernel<CNT>, wfTraceLeanKernel<BW, CP, CNT, COOP, PH>, wfTraceLeanStreamKernel<CP, CNT>,
    wfTraceSwarmKernel<CNT, COOP>, wfTraceGroupKernel<G, CAPX, CNT>, wfGenerateKernel<CNT>, wfShadeKernel<POOL, CNT>."""
    if "<" not in kernel:
        return False
    name, args = kernel.split("<", 1)
    args = [a.strip() for a in args.rstrip("> ").split(",")]
    at = {"wfTraceKernel": 0, "wfTraceLeanKernel": 2, "wfTraceLeanStreamKernel": 1, "wfTraceSwarmKernel": 0, "wfTraceGroupKernel": 2,
          "wfGenerateKernel": 0, "wfShadeKernel": 1}.get(name.split("::")[-1])
    return at is not None and at < len(args) and args[at] == "true"


per = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
meta = {}
for f in sorted(glob.glob(f"gpurun_out/pmc_{pmc_tag}_*/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wf" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        per[short][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[short][r["Counter_Name"]] += 1
        meta[short] = {x: r[x] for x in ("Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
out = {"note": note, "frames_profiled": frames, "fetch_size_factor": factor, "per_frame": {}}
trace = collections.defaultdict(float)
for k, c in sorted(per.items()):
    counting = is_counting(k)                      # CNT instantiations: the un-timed counting renders of bench.py
    out["per_frame"][k] = {"dispatch": meta[k], "counting_render": counting, "launches_per_frame": max(calls[k].values()) / frames,
                           "counters": {n: v / frames for n, v in sorted(c.items())}}
    if "Trace" in k and not counting:
        for n, v in c.items():
            trace[n] += v / frames
d = {}
if "SQ_INSTS_VALU" in trace:
    d["instructions_per_frame"] = {n: trace[n] for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS") if n in trace}
if "SQ_BUSY_CYCLES" in trace and "SQ_ACTIVE_INST_VALU" in trace:
    d["sq"] = {n: trace[n] for n in trace if n.startswith("SQ_")}
if "TCC_EA0_RDREQ_sum" in trace:
    d["hbm_read_requests_per_frame"] = trace["TCC_EA0_RDREQ_sum"]
    d["hbm_read_requests_128B_per_frame"] = trace.get("TCC_EA0_RDREQ_128B_sum")
if "TCC_HIT_sum" in trace:
    d["l2_hit_rate"] = trace["TCC_HIT_sum"] / max(1.0, trace["TCC_HIT_sum"] + trace["TCC_MISS_sum"])
if "TCP_TCC_READ_REQ_sum" in trace:
    d["l1_hit_rate"] = 1 - trace["TCP_TCC_READ_REQ_sum"] / max(1.0, trace["TCP_TOTAL_CACHE_ACCESSES_sum"])
if "FETCH_SIZE" in trace:
    d["hbm_read_bytes_per_frame"] = int(trace["FETCH_SIZE"] * 1024 * factor)
    d["hbm_write_bytes_per_frame"] = int(trace.get("WRITE_SIZE", 0) * 1024)
    d["hbm_bytes_per_frame"] = d["hbm_read_bytes_per_frame"] + d["hbm_write_bytes_per_frame"]
    d["by_kernel"] = {k: int(v["counters"].get("FETCH_SIZE", 0) * 1024 * factor + v["counters"].get("WRITE_SIZE", 0) * 1024)
                      for k, v in out["per_frame"].items() if not v["counting_render"]}
out["trace_kernels_per_frame"] = d
json.dump(out, open(f"profiles/{name}_pmc.json", "w"), indent=1)
print(json.dumps(d, indent=1))
if "hbm_bytes_per_frame" in d and traffic_file != "-":
    wl = json.loads(match)
    wl["hbm_bytes_per_frame"] = d["hbm_bytes_per_frame"]
    wl["source"] = f"profiles/{name}_pmc.json"
    wl["csrc_sha256"] = bench.csrc_hash()
    wl["tuning"] = tuning
    json.dump(wl, open(f"profiles/{traffic_file}", "w"), indent=1)
