"""Diagnostic: where does a device-built tree give another primary id than the golden (ties?)"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from conftest import load_golden
from raytracing_amd import abi, scenes
from raytracing_amd.tracer import Tracer
name = sys.argv[1] if len(sys.argv) > 1 else "bvh_viewB_256"
g = load_golden(name)
m = scenes.bvh_default()
t = Tracer(0)
v = m.scene_view(); v.bvh = None; v.bvhNodeNum = 0
t.scene_upload(v); t.set_camera(g["trans"], g["width"], g["height"]); t.set_decay_origin((0, 0, 0, 1))
img, hits = t.render(max_depth=g["depth"], mode=abi.MODE_PARITY)
t.scene_upload(m.scene_view())
img0, hits0 = t.render(max_depth=g["depth"], mode=abi.MODE_PARITY)
bad = ~(hits == hits0).all(axis=-1)
d = np.abs(img - img0).max(axis=-1)
print("ids differing", bad.sum(), "pixels differing > 1e-3", (d > 1e-3).sum(), "both", (bad & (d > 1e-3)).sum())
for y, x in list(zip(*np.nonzero(bad | (d > 1e-3))))[:24]:
    print("  px", x, y, "host", hits0[y, x], "device", hits[y, x], "golden", g["ids"][y, x], "t", g["t"][y, x], "diff", d[y, x])
