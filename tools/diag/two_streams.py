"""Feasibility probe: do two wavefront pipelines on two streams overlap (shade is HBM-bound, trace VALU-bound)?
Two contexts on one GPU render half of C3's / C4's samples each, from two host threads, against one context rendering all of them."""
import sys, os, time, threading
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from raytracing_amd import scenes, abi, host, tracer as T
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
if which == "c3":
    m = scenes.pool(); W, H, depth, spp, variant = 1920, 1080, 8, 64, abi.VARIANT_POOL
else:
    m = scenes.mesh_scene(subdivisions=2, builder="reference"); W, H, depth, spp, variant = 3840, 2160, 4, 64, abi.VARIANT_BVH
m.camera_set_default(H) if hasattr(m, "camera_set_default") else None
tr = m.camera(H)
jit = host.jitter_table(spp)
dev = torch.device("cuda:0")
def make():
    t = T.Tracer(0); t.scene_upload(m.scene_view()); t.set_camera(tr, W, H); t.set_decay_origin((0, 0, 0, 1))
    return t, torch.zeros((H, W, 4), dtype=torch.float32, device=dev), torch.cuda.Stream(device=dev)
ctx = [make(), make()]
kw = dict(max_depth=depth, variant=variant, mode=abi.MODE_FAST, jitter=jit)
def run(i, lo, hi, pin):
    t, out, st = ctx[i]
    t.set_tuning(phasedTrips=pin)
    t.render_device(out.data_ptr(), stream=st.cuda_stream, samples=(lo, hi), **kw)
def frame(split, pin):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if split:
        th = [threading.Thread(target=run, args=(i, i * spp // 2, (i + 1) * spp // 2, pin)) for i in range(2)]
        for x in th: x.start()
        for x in th: x.join()
    else:
        run(0, 0, spp, pin)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for pin in (1, 2):
    for split in (False, True, False, True):
        for _ in range(4): frame(split, pin)
        ts = [frame(split, pin) for _ in range(4)]
        print(which, "phasedTrips", pin, "two contexts" if split else "one context ", "ms/frame", round(min(ts), 2), [round(x, 1) for x in ts], flush=True)
