"""Diagnostic: GPU PARITY vs oracle vs llvmpipe golden on a Pool golden (which pixels / ids differ)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_golden
from oracle import oracle_api as O
from raytracing_amd import abi, scenes
from raytracing_amd.tracer import Tracer

name = sys.argv[1] if len(sys.argv) > 1 else "pool_viewA_160"
g = load_golden(name)
m = scenes.pool()
view = m.scene_view()
oimg, ohits, _ = O.render(view, g["trans"], g["width"], g["height"], max_depth=g["depth"], variant=g["variant"], threads=8)
t = Tracer(0)
t.scene_upload(view)
t.set_camera(g["trans"], g["width"], g["height"])
t.set_decay_origin((0, 0, 0, 1))
for mega, mode in ((False, abi.MODE_PARITY), (True, abi.MODE_PARITY), (False, abi.MODE_FAST), (True, abi.MODE_FAST)):
    img, hits = t.render(max_depth=g["depth"], variant=g["variant"], mode=mode, megakernel=mega)
    dg = ~(hits == g["ids"]).all(axis=-1)
    do = ~(hits == ohits).all(axis=-1)
    og = ~(ohits == g["ids"]).all(axis=-1)
    print("mega", mega, "mode", mode, "gpu!=gold", dg.sum(), "gpu!=oracle", do.sum(), "oracle!=gold", og.sum(),
          "img gpu==oracle bits", np.array_equal(img.view(np.uint32), oimg.view(np.uint32)), np.abs(img - oimg).max())
    for y, x in zip(*np.nonzero(do)):
        print("  px", x, y, "gpu", hits[y, x], "oracle", ohits[y, x], "gold", g["ids"][y, x])
