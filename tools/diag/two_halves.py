"""Feasibility probe: a 1-spp frame (C2) as two half frames (interleaved 16-row stripes) on two contexts and two streams, so that the tail
of one pipeline's launches (a launch ends with its longest rays) overlaps the other's dense phase.  Against one context rendering the
whole frame.  Times N frames back to back (the host queues ahead; one synchronise at the end)."""
import sys, os, time, threading
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from raytracing_amd import scenes, abi, tracer as T
m = scenes.mesh_scene(subdivisions=2, builder="reference"); W, H, depth = 1920, 1080, 4
tr = m.camera(H)
dev = torch.device("cuda:0")
def make():
    t = T.Tracer(0); t.scene_upload(m.scene_view()); t.set_camera(tr, W, H); t.set_decay_origin((0, 0, 0, 1))
    t.set_tuning(phasedTrips=2)
    return t, torch.zeros((H, W, 4), dtype=torch.float32, device=dev), torch.cuda.Stream(device=dev)
ctx = [make(), make()]
kw = dict(max_depth=depth, mode=abi.MODE_FAST)
N = 40
def loop(i, rows):
    t, out, st = ctx[i]
    for _ in range(N):
        t.render_device(out.data_ptr(), stream=st.cuda_stream, rows=rows, **kw)
def timed(split):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if split:
        th = [threading.Thread(target=loop, args=(i, ("interleave", 2, i))) for i in range(2)]
        for x in th: x.start()
        for x in th: x.join()
    else:
        loop(0, None)
    for t, _, _ in ctx: t.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / N
for split in (False, True, False, True):
    timed(split)
    ts = [timed(split) for _ in range(3)]
    print("two half frames on two streams" if split else "one frame on one stream       ", "ms/frame", round(min(ts), 4), [round(x, 3) for x in ts], flush=True)
