"""The JSON line bench.py prints, against the driver's contract: checked on the lines the round recorded under profiles/ (CPU) and on a
live one-rank run (GPU).  A missing key or a wrong type is found here rather than by the driver."""
import glob
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

TOP = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": (int, float),
       "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}
ROOFLINE = {"bound": str, "achieved": (int, float), "peak": (int, float), "unit": str}
CPU = {"value": (int, float), "unit": str, "cores": int, "kind": str, "sample": str}


def check_line(line, default_run):
    for k, t in TOP.items():
        assert k in line and isinstance(line[k], t), (k, line.get(k))
    assert "vs_baseline" in line and line["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert line["higher_is_better"] is True and line["scaling"] in ("weak", "strong") and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert isinstance(line["config"].get("workload"), str) and "model" not in line["config"]
    assert line["value"] > 0 and line["ms_per_step"] > 0
    r = line["roofline"]
    for k, t in ROOFLINE.items():
        assert k in r and isinstance(r[k], t), (k, r.get(k))
    assert "frac" in r and "traffic" in r
    if r["frac"] is not None:
        assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3 * max(1.0, r["frac"])
    else:
        assert r.get("profile_matches_build") is False and "stale_frac" in r          # withheld, and said why
    if default_run:
        c = line["cpu_baseline"]
        for k, t in CPU.items():
            assert k in c and isinstance(c[k], t), (k, c.get(k))
        assert c["kind"] in ("port", "reference")
        c5 = line["roofline_c5"]
        assert c5["bound"] == "hbm" and c5["peak"] == 8000.0 and c5["unit"] == "GB/s" and "verified" in c5
        assert c5["verified"]["ok"] is True and c5["verified"]["pixels"] >= 256


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(REPO, "profiles", "r03_bench_*.json"))), ids=os.path.basename)
def test_recorded_bench_lines_keep_the_contract(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "one JSON line per run"
    check_line(json.loads(lines[0]), default_run=path.endswith("r03_bench_default.json"))


@pytest.mark.gpu
def test_live_bench_line_keeps_the_contract():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "2", "--no-c5", "--cpu-rows", "64"], capture_output=True, text=True,
                       timeout=400, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip() == lines[0], "stdout is the one JSON line"
    line = json.loads(lines[0])
    check_line(line, default_run=False)
    assert line["steps"] == 3 and line["warmup"] == 2 and line["n_gpus"] == 1 and "cpu_baseline" in line
