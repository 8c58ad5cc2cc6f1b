#!/usr/bin/env python3
"""Reads a per-wave timeline dumped by liblrltrace.so under LRL_TIMELINE=<file> (diagnostic builds of the trace kernel:
{start, end} in 100 MHz s_memrealtime ticks + hardware id per wave) and prints the launch's wave-slot occupancy over time.
usage: tools/timeline.py file [bins]"""
import sys
import numpy as np
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 20
words = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, words)
if words == 6:
    a = a[:131072]          # the lean kernel's waves; the hand-over kernel's entries follow (tools/swarm_stats.py)
a = a[a[:, 1] > 0]
t0, t1 = a[:, 0].min(), a[:, 1].max()
s, e = (a[:, 0] - t0).astype(np.float64) / 100.0, (a[:, 1] - t0).astype(np.float64) / 100.0   # microseconds
span = e.max()
d = e - s
hw = a[:, 2] & 0xffffffff
xcc = (a[:, 2] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print(f"waves {len(a)}  span {span:.1f} us  wave duration us: mean {d.mean():.1f} median {np.median(d):.1f} p90 {np.percentile(d,90):.1f} max {d.max():.1f}")
print(f"mean resident waves {d.sum()/span:.0f}  (per CU of 256: {d.sum()/span/256:.2f})")
edges = np.linspace(0, span, bins + 1)
for i in range(bins):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(e, hi) - np.maximum(s, lo), 0, None).sum() / (hi - lo)
    started = ((s >= lo) & (s < hi)).sum()
    print(f"  {lo:8.1f}-{hi:8.1f} us  resident {ov:7.0f}  started {started}")
ids = xcc * 4096 + se * 512 + sh * 256 + cu
u = np.unique(ids)
print(f"distinct (xcc,se,sh,cu) ids seen: {len(u)}; xcc histogram {np.bincount(xcc.astype(int))}")
if words >= 6:
    mx, sm, it = a[:, 3].astype(np.float64), a[:, 4].astype(np.float64), a[:, 5].astype(np.float64)
    print(f"trips of one lane-path/ray: chip max {mx.max():.0f}; per wave: loop trips mean {it.mean():.0f} max {it.max():.0f}; lane efficiency sum/(64*trips) {sm.sum()/(64*it.sum()):.3f}")
    k = np.argsort(-d)[:8]
    for i in k: print(f"   wave {i}: {d[i]:.0f} us, trips {it[i]:.0f}, max lane steps {mx[i]:.0f}, sum {sm[i]:.0f}  -> {d[i]*1e3/max(it[i],1):.0f} ns/trip")
    print(f"all waves: ns per loop trip {d.sum()*1e3/it.sum():.0f}")
