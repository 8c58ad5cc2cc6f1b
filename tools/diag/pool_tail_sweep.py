"""One-sample 1080p Pool frames (the demo's own loop) with the tail launch starting at different segments (LrlTuning.tailFrom)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from raytracing_amd import scenes, abi, host, tracer as T
m = scenes.pool()
W, H = 1920, 1080
t = T.Tracer(0)
t.scene_upload(m.scene_view()); t.set_camera(m.camera(H), W, H); t.set_decay_origin((0, 0, 0, 1))
out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
def frame():
    t.render_device(out.data_ptr(), max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST)
for tail in (0xffffffff, 3, 4, 5, 6, 8):
    t.set_tuning(tailFrom=tail, phasedTrips=1)
    for _ in range(10): frame()
    t.sync(); t0 = time.perf_counter()
    for _ in range(100): frame()
    t.sync()
    print("tailFrom", tail if tail < 100 else "never", round((time.perf_counter() - t0) * 10, 4), "ms per one-sample 1080p Pool frame")
