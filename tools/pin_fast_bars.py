#!/usr/bin/env python3
"""After a run of `pytest -m gpu` on the GPU box: gpurun_out/observed_bars.json (what every image / id check observed, tests/conftest.py)
-> profiles/<round>_observed_bars.json (tracked, merged: entries of checks that did not run this time are kept) and
tests/golden/fast_bars.json (the per-check pins of the FAST image bars: observed share of pixels that moved by more than 1e-3 and
trimmed RMSE; conftest.check_image holds a pinned check to 1.5 x these).
usage: tools/pin_fast_bars.py r03 [--pin]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
obs = json.load(open(os.path.join(REPO, "gpurun_out", "observed_bars.json")))
dst = os.path.join(REPO, "profiles", f"{rnd}_observed_bars.json")
try:
    old = json.load(open(dst))
except OSError:
    old = {}
old.update(obs)
json.dump(old, open(dst, "w"), indent=0, sort_keys=True)
print(len(obs), "observed,", len(old), "kept in", dst)
if "--pin" in sys.argv:
    pins = {}
    for k, st in old.items():
        what, mode = k.rsplit("|", 1)
        if mode == "1" and "frac_gt_1e-3" in st:
            pins[what] = {"frac_gt_1e-3": st["frac_gt_1e-3"], "rmse_trim": st["rmse_trim"], "rmse": st["rmse"]}
    json.dump(pins, open(os.path.join(REPO, "tests", "golden", "fast_bars.json"), "w"), indent=0, sort_keys=True)
    print(len(pins), "FAST checks pinned")
