#!/bin/bash
# here, after tools/r03_final.sh c2 / c5 ran on the GPU box: fold their outputs into profiles/ (the traffic files get the sources' hash)
set -e
python3 tools/summarize_round.py r03_c2 r03_c2_stats c2 19 r03_traffic.json '{"workload_triangles": 1240960, "mode": "fast", "builder": "reference", "width": 1920, "height": 1080, "spp": 1, "depth": 4, "diffuse": false}' "C2, round 3: FAST wavefront pipeline, caller's tree packed, fitted and laid out on the device; 8-B hit records, 16-B shadow rays (19 frames a pass: bench.py --steps 4 --warmup 10 --tuning phasedTrips=2 and its five per-launch timing frames)" "phasedTrips=2" > /dev/null
python3 tools/summarize_round.py r03_c3 r03_c3_stats c3 11 - '{}' "C3, round 3 (SQ counters only): bench.py --workload c3 --steps 2 --warmup 8 --tuning phasedTrips=1" "phasedTrips=1" > /dev/null
python3 tools/summarize_round.py r03_c5 r03_c5_stats c5 17 r03_traffic_c5.json '{"workload_triangles": 100517760, "spp": 16, "depth": 4}' "C5 (100.5 M triangles, PLOC tree built on the device, four-wide nodes and triangle records laid out in the order the rays walk them, diffuse bounces, 16 spp), round 3 final kernels: bench.py --workload c5 --steps 4 --warmup 8 --tuning phasedTrips=2 (17 frames a pass)" "phasedTrips=2" > /dev/null
python3 - <<'PY'
import json
for n in ("r03_traffic.json", "r03_traffic_c5.json"):
    d = json.load(open("profiles/" + n)); print(n, d["hbm_bytes_per_frame"], d["csrc_sha256"][:12])
import bench; print("sources now", bench.csrc_hash()[:12])
for n, per in (("r03_c2", 5), ("r03_c5", 5)):
    d = json.load(open(f"profiles/{n}_pmc.json"))
    for k, v in d["per_frame"].items():
        if "Lean" in k and not v["counting_render"]: print(n, k[-40:], v["launches_per_frame"])
PY
