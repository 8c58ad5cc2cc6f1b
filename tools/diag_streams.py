"""Diagnostic: (1) render on a torch side stream; (2) under torchrun + gloo: async all_reduce of a CUDA tensor from a side stream."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import abi, scenes, tracer
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
m = scenes.mesh_scene(subdivisions=0, builder="reference")
W, H = 640, 360
t = tracer.Tracer(0); t.scene_upload(m.scene_view()); t.set_camera(m.camera(H), W, H); t.set_decay_origin((0, 0, 0, 1))
out = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    t.render_device(out.data_ptr(), stream=s.cuda_stream, max_depth=4, mode=abi.MODE_FAST)
torch.cuda.synchronize()
print("side-stream render done, mean", float(out.mean()), flush=True)
if "RANK" in os.environ:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    x = torch.ones(1000, device=dev)
    w = dist.all_reduce(x, async_op=True); w.wait(); torch.cuda.synchronize()
    print("default-stream async all_reduce ok", float(x[0]), flush=True)
    with torch.cuda.stream(s):
        t.render_device(out.data_ptr(), stream=s.cuda_stream, max_depth=4, mode=abi.MODE_FAST)
        w = dist.all_reduce(out, async_op=True)
    print("enqueued", flush=True)
    w.wait(); torch.cuda.synchronize()
    print("side-stream render + async all_reduce ok", float(out.mean()), flush=True)
    dist.barrier(); dist.destroy_process_group()
