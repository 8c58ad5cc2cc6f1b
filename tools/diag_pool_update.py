"""Diagnostic: Pool scene with lowered water, PARITY vs FAST (and LRL_TRACE=ref FAST) after lrl_scene_update_triangles."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from conftest import load_golden
from raytracing_amd import abi, scenes
from raytracing_amd.tracer import Tracer
g = load_golden("pool_viewB_192")
m = scenes.pool()
t = Tracer(0)
t.scene_upload(m.scene_view()); t.set_camera(g["trans"], g["width"], g["height"]); t.set_decay_origin((0, 0, 0, 1))
water = m.arrays()["triangles"][: 2 * 47 * 47].copy()
moved = water.copy()
for f in ("p1", "p2", "p3"): moved[f][:, 2] -= 0.05
print("water z range", min(water[f][:, 2].min() for f in ("p1", "p2", "p3")), max(water[f][:, 2].max() for f in ("p1", "p2", "p3")))
t.update_triangles(0, moved)
low, lh = t.render(max_depth=8, variant=abi.VARIANT_POOL)
for depth in (1, 8):
    p, ph = t.render(max_depth=depth, variant=abi.VARIANT_POOL)
    f, fh = t.render(max_depth=depth, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST)
    d = np.abs(p - f).max(axis=-1)
    bad = d > 1e-3
    print("depth", depth, "pixels differing", bad.sum(), "ids differing", (~(ph == fh).all(axis=-1)).sum(), "max", d.max())
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:12]:
        print("   px", x, y, "parity id", ph[y, x], "fast id", fh[y, x], "parity", p[y, x, :3], "fast", f[y, x, :3])
