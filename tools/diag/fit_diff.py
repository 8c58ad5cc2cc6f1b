import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from raytracing_amd import scenes, abi, host, tracer as T
from conftest import load_golden
m = scenes.pool(); tris = m.arrays()["triangles"]
gx, gy = scenes.POOL_GRID; k, dt = scenes.POOL_WATER
g = load_golden("pool_viewB_192"); W, H = 192, 192
for spp in (1, 8, 64):
    jit = host.jitter_table(spp)
    kw = dict(max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST, jitter=jit)
    t = T.Tracer(0)
    t.scene_upload(m.scene_view()); t.set_camera(g["trans"], W, H); t.set_decay_origin((0, 0, 0, 1))
    t.water_init(gx, gy, k, dt, scenes.pool_water_points(m), 0, tris[:2 * (gx - 1) * (gy - 1)])
    for steps in (0, 200):
        t.water_step(steps)
        loose = t.render(**kw); assert t.scene_info()["ownBoxes"] == 0
        for i in range(400):
            fit = t.render(**kw)
            if t.scene_info()["ownBoxes"]: break
        fit = t.render(**kw)
        d = (loose[0].view(np.uint32) != fit[0].view(np.uint32)).any(axis=-1)
        print("spp", spp, "steps", steps, "renders", i, "pixels differing", int(d.sum()), "max abs", float(np.abs(loose[0] - fit[0]).max()), np.argwhere(d)[:5].tolist(), flush=True)
    t.close()
