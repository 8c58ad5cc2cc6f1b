#!/usr/bin/env python3
"""Experiment: one frame rendered as K row stripes by K contexts (own streams) from K host threads, vs one context."""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np, torch
from raytracing_amd import abi, scenes, tracer
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 1920, 1080
m = scenes.mesh_scene(subdivisions=2, builder="reference")
trans = m.camera(H); view = m.scene_view()
trs = []
for k in range(K):
    t = tracer.Tracer(0); t.scene_upload(view); t.set_camera(trans, W, H); t.set_decay_origin((0, 0, 0, 1)); trs.append(t)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
# stripes of 8-row granularity, interleaved in blocks so that every context gets a share of the heavy middle
rows8 = H // 8
bounds = [(i * 8, min(H, (i + 1) * 8)) for i in range(rows8)]
def work(k, reps):
    t = trs[k]
    for _ in range(reps):
        for i in range(k, rows8, K * STRIDE) if False else range(0):
            pass
def frame_single():
    trs[0].render_device(out.data_ptr(), max_depth=4, mode=abi.MODE_FAST); trs[0].sync()
def frame_multi(S):
    # S contiguous stripes, context k takes stripes k, k+K, ...
    edges = [round(i * H / S / 8) * 8 for i in range(S)] + [H]
    def run(k):
        for s in range(k, S, K):
            r0, r1 = edges[s], edges[s + 1]
            if r1 > r0:
                trs[k].render_device(out.data_ptr() + r0 * W * 16, max_depth=4, mode=abi.MODE_FAST, rows=(r0, r1))
        trs[k].sync()
    th = [threading.Thread(target=run, args=(k,)) for k in range(K)]
    [t.start() for t in th]; [t.join() for t in th]
for name, fn in [("single", frame_single)] + [(f"K={K} stripes={S}", (lambda S=S: frame_multi(S))) for S in (K, 2 * K, 4 * K)]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt * 1e3:.3f} ms per frame")
