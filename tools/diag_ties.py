"""Diagnostic: how much of a FAST frame depends on the schedule (lean kernel only / hand-over after 4 trips with 16 or 64 lanes per ray /
defaults), on the views whose FAST bars are tightest.  Prints pairwise id mismatches, image RMSE and the mismatching pixels' hits."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden, image_stats
from raytracing_amd import abi, scenes, tracer as T

t = T.Tracer(0)
base = t.get_tuning()
for scene, name in (("pool", "pool_viewA_160"), ("pool", "pool_viewB_192"), ("default", "bvh_viewB_256")):
    m = scenes.pool() if scene == "pool" else scenes.bvh_default()
    g = load_golden(name)
    res = {}
    for label, kw in (("default", {}), ("lean-only", dict(handoverEarly=0, handoverLate=0, adapt=0)), ("group16@4", dict(adapt=0, handoverList=[4], groupLanes=16)),
                      ("wave64@4", dict(adapt=0, handoverList=[4], groupLanes=64))):
        t.set_tuning(base)
        if kw:
            t.set_tuning(**kw)
        t.scene_upload(m.scene_view())
        t.set_camera(g["trans"], g["width"], g["height"])
        t.set_decay_origin((0, 0, 0, 1))
        res[label] = t.render(max_depth=g["depth"], variant=g["variant"], mode=abi.MODE_FAST)
    t.set_tuning(base)
    par = t.render(max_depth=g["depth"], variant=g["variant"], mode=abi.MODE_PARITY)
    print(f"== {name}")
    for label, (img, hits) in res.items():
        st = image_stats(img, g["rgb"])
        sp = image_stats(img, par[0])
        print(f"  {label:10s} vs golden: ids differ {int((hits != g['ids']).any(axis=-1).sum()):4d}  rmse {st['rmse']:.5f} max {st['max']:.3f} frac {st['frac_gt_1e-3']:.5f} | vs PARITY rmse {sp['rmse']:.5f}")
    labels = list(res)
    for i in range(len(labels)):
        for j in range(i + 1, len(labels)):
            a, b = res[labels[i]], res[labels[j]]
            d = np.abs(a[0][..., :3] - b[0][..., :3]).max(axis=-1)
            print(f"  {labels[i]:10s} vs {labels[j]:10s}: ids differ {int((a[1] != b[1]).any(axis=-1).sum()):4d}, pixels differ {int((d > 0).sum()):5d}, > 1e-3: {int((d > 1e-3).sum()):4d}, max {d.max():.3f}")
    a, b = res["lean-only"], res["group16@4"]
    d = np.abs(a[0][..., :3] - b[0][..., :3]).max(axis=-1)
    ys, xs = np.nonzero(d > 1e-3)
    for y, x in list(zip(ys, xs))[:12]:
        print(f"    pixel ({x},{y}): lean {a[1][y, x]} {a[0][y, x, :3]}  group {b[1][y, x]} {b[0][y, x, :3]}  golden {g['ids'][y, x]} {g['rgb'][y, x]}")
t.close()
