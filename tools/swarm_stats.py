#!/usr/bin/env python3
"""Reads the wave-per-ray kernel's part of a timeline dump (diagnostic build, entries 131072...): per handed-over ray
{start, end, refs handed over, rounds, entries taken, largest frontier}.  usage: tools/swarm_stats.py file"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 6)[131072:]
a = a[a[:, 1] > 0]
if len(a) == 0:
    print("no wave-per-ray entries"); sys.exit(0)
d = (a[:, 1] - a[:, 0]).astype(np.float64) / 100.0
start, rounds, taken, mx = a[:, 2].astype(float), a[:, 3].astype(float), a[:, 4].astype(float), a[:, 5].astype(float)
span = (a[:, 1].max() - a[:, 0].min()) / 100.0
print(f"rays {len(a)}  span {span:.1f} us  wave us: mean {d.mean():.1f} p50 {np.median(d):.1f} p90 {np.percentile(d, 90):.1f} max {d.max():.1f}")
print(f"refs handed over: mean {start.mean():.1f} max {start.max():.0f};  rounds: mean {rounds.mean():.1f} p90 {np.percentile(rounds, 90):.0f} max {rounds.max():.0f}")
print(f"entries taken per round: mean {taken.sum() / rounds.sum():.1f};  per ray: mean {taken.mean():.0f} max {taken.max():.0f};  largest frontier: mean {mx.mean():.0f} max {mx.max():.0f}")
print(f"us per round: {d.sum() / rounds.sum():.2f}")
