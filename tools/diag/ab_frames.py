"""A/B of two builds of the library, frame by frame: renders a few FAST frames (default scene with its glass, the Pool with sample batches, a bunny with
diffuse bounces, a sharded frame) and writes them to an .npz; run once per library (env LRLTRACE_LIB picks it), then `cmp` mode compares.
usage: ab_frames.py dump out.npz [tuning=value,...] | ab_frames.py cmp a.npz b.npz"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np


def dump(path, tuning=""):
    import torch
    tune = {k: (float(v) if "." in v else int(v)) for k, v in (kv.split("=") for kv in tuning.split(",") if kv)}
    from raytracing_amd import scenes, abi, host, tracer as T
    out = {}

    def frames(name, m, W, H, spp, depth, variant, diffuse=False, rows=None):
        t = T.Tracer(0)
        if tune:
            t.set_tuning(**tune)
        t.scene_upload(m.scene_view()); t.set_camera(m.camera(H), W, H); t.set_decay_origin((0, 0, 0, 1))
        buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        for i in range(3):      # (the schedule adapts over the first frames; the image must not)
            buf.zero_()
            t.render_device(buf.data_ptr(), max_depth=depth, variant=variant, mode=abi.MODE_FAST, jitter=host.jitter_table(spp) if spp > 1 else None, diffuse=diffuse, rows=rows)
            t.sync()
            out[f"{name}_{i}"] = buf.cpu().numpy().copy()

    frames("default", scenes.bvh_default(), 640, 480, 1, 10, abi.VARIANT_BVH)
    frames("default4", scenes.bvh_default(), 320, 240, 4, 10, abi.VARIANT_BVH)
    frames("default_rows", scenes.bvh_default(), 320, 240, 1, 10, abi.VARIANT_BVH, rows=(64, 160))
    frames("pool", scenes.pool(), 480, 270, 8, 8, abi.VARIANT_POOL)
    b = scenes.mesh_scene(1)
    frames("bunny", b, 640, 360, 1, 4, abi.VARIANT_BVH)
    frames("bunny_diffuse", b, 640, 360, 2, 4, abi.VARIANT_BVH, diffuse=True)
    np.savez_compressed(path, **out)
    print("wrote", path, sorted(out))


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in sorted(A.files):
        same = np.array_equal(A[k].view(np.uint32), B[k].view(np.uint32))
        n = int((A[k].view(np.uint32) != B[k].view(np.uint32)).any(axis=-1).sum())
        print(k, "identical" if same else f"DIFFERENT in {n} pixels, max abs {np.abs(A[k] - B[k]).max():.3g}")
        bad += 0 if same else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    dump(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "") if sys.argv[1] == "dump" else cmp(sys.argv[2], sys.argv[3])
