import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from raytracing_amd import scenes, abi, host, tracer as T
from oracle import oracle_api as O
from conftest import load_golden
m = scenes.pool(); tris = m.arrays()["triangles"]
g = load_golden("pool_viewB_192"); W, H = 192, 192
jit = host.jitter_table(64)
y, x = 8, 113
res = {}
for how in ("fitted", "loose"):
    t = T.Tracer(0)
    t.scene_upload(m.scene_view()); t.set_camera(g["trans"], W, H); t.set_decay_origin((0, 0, 0, 1))
    if how == "loose":
        mv = tris[:1].copy(); mv["p1"][:, 2] -= 0.01
        t.update_triangles(0, mv); t.update_triangles(0, tris[:1])
    print(how, t.scene_info()["ownBoxes"])
    out = []
    for s in range(64):
        for mode in (abi.MODE_FAST, abi.MODE_PARITY):
            img, hits = t.render(max_depth=8, variant=abi.VARIANT_POOL, mode=mode, jitter=jit, samples=(s, s + 1))
            out.append((img[y, x].copy(), hits[y, x].copy()))
    res[how] = out
    t.close()
ts = host.sample_trans(g["trans"], jit)
for s in range(64):
    f, l = res["fitted"][2 * s], res["loose"][2 * s]
    if not np.array_equal(f[0].view(np.uint32), l[0].view(np.uint32)):
        par = res["loose"][2 * s + 1]
        oimg, ohits, _ = O.render(m.scene_view(), ts[s:s + 1], W, H, max_depth=8, variant=abi.VARIANT_POOL, threads=8)
        print("sample", s, "fitted", f[0] * 64, f[1], "loose", l[0] * 64, l[1], "parity", par[0] * 64, par[1], "oracle", oimg[y, x], ohits[y, x])
