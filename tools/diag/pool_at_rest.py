"""The Pool's binding flow (scene upload, lrl_water_init, lrl_water_step(0), 64-spp frames, then water substeps before every frame, then the
demo's own loop of one-sample frames with a step each): frame times with the caller's four-wide tree refitted on the device behind every
step (lrl_refit_node4)."""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from raytracing_amd import scenes, abi, host, tracer as T
m = scenes.pool()
W, H, spp = 1920, 1080, 64
gx, gy = scenes.POOL_GRID
k, dt = scenes.POOL_WATER
tris = m.arrays()["triangles"]
t = T.Tracer(0)
t.scene_upload(m.scene_view()); t.set_camera(m.camera(H), W, H); t.set_decay_origin((0, 0, 0, 1))
out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
jit = host.jitter_table(spp)
def frame(j=jit, steps=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if steps is not None: t.water_step(steps)
    t.render_device(out.data_ptr(), max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST, jitter=j)
    t.sync(); return (time.perf_counter() - t0) * 1e3
t.set_tuning(phasedTrips=1)
for _ in range(4): frame()
print("as uploaded                 ", round(frame(), 2), "ms, own boxes", t.scene_info()["ownBoxes"])
t.water_init(gx, gy, k, dt, scenes.pool_water_points(m), 0, tris[:2 * (gx - 1) * (gy - 1)])
print("with lrl_water_step(0)      ", round(frame(steps=0), 2), "ms, own boxes", t.scene_info()["ownBoxes"])
print("next frame                  ", round(frame(), 2), "ms")
for i in range(3):
    print("200 substeps + 64-spp frame ", round(frame(steps=200), 2), "ms, own boxes", t.scene_info()["ownBoxes"])
for _ in range(8): frame(None, 1)
one = [frame(None, None) for _ in range(20)]
step = [frame(None, 1) for _ in range(20)]
print("one-sample frame            ", round(sorted(one)[10], 3), "ms;  with a water substep, TrianglePre, boxes and the refit before it", round(sorted(step)[10], 3), "ms")
