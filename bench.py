#!/usr/bin/env python3
"""bench.py — Mrays/s of the hot path on the BASELINE.json headline configuration.

Workload (BASELINE.json configs[1]): Stanford bunny x 2 midpoint subdivisions
(1 240 960 triangles) + ground disc + 1 point light, 1920x1080, 1 spp, 4 bounces.  A "step" is one
frame: the wavefront pipeline (generate, then trace + shade per ray segment) over every pixel, scene
and camera already resident in HBM, image written to HBM.  rays = primary + reflection + refraction
+ shadow segments (device counters of an un-timed counting render, which walks the tree in the
reference's order; checked against the CPU oracle in tests/).

Roofline: the dominant kernels are the trace kernels (one launch per ray segment: 5 per frame here).
achieved = algorithmic bytes of one frame / device time of that frame's trace launches, timed with
HIP events around every trace launch in a separate short pass (the event records add gaps, so that
pass is not the one `value` comes from).  frame_ms is the device time of the whole pipeline.

N > 1 (torchrun, one rank per GPU): the scene is replicated; every rank renders the whole frame for
its own sample(s) of an N x spp jitter table (sample-index sharding, so per-GPU work is fixed: weak
scaling) and the partial means are summed with one RCCL all-reduce per step, inside the timed
region; the all-reduce of a frame overlaps the rendering of the next one (two image buffers).
value = rays of all ranks / max-over-ranks time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fast|parity] [--builder reference|median] [--workload c2|c3]

--workload c3 measures SURVEY 8d's C3 (Pool demo scene, Pool shader variant, 64 spp, depth 8) the same way; it is not the
headline line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="fast", choices=["fast", "parity"])
    ap.add_argument("--builder", default="reference", choices=["reference", "median", "device"],
                    help="who builds the BVH: the restated reference builder (host, default), a median-split host builder, or the library on the device")
    ap.add_argument("--subdivisions", type=int, default=2)
    ap.add_argument("--width", type=int, default=0, help="0 = the workload's own: 1920 (c1: 1024)")
    ap.add_argument("--height", type=int, default=0, help="0 = the workload's own: 1080 (c1: 1024)")
    ap.add_argument("--depth", type=int, default=0, help="RayTraceDepth (0 = the workload's own: 4 for c2, 8 for c3)")
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3"],
                    help="c2 (default, the headline metric's configuration) | c1: the reference's default BVH scene as shipped, 1024 x 1024, "
                         "RayTraceDepth 10 | c3: Pool demo scene, Pool shader variant, 64 spp (SURVEY 8d)")
    ap.add_argument("--grid", type=int, default=1, help="c2 only: grid x grid replicated copies of the mesh (towards C5: 9 -> 100.5 M triangles)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (0 = the workload's own: 1 for c2, 64 for c3)")
    ap.add_argument("--diffuse", action="store_true", help="LRL_FLAG_DIFFUSE_BOUNCE: cosine-weighted continuation rays (SURVEY 8d C5: incoherent secondary rays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline renders (0 = auto)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from raytracing_amd import abi, host, scenes, tracer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal on a one-GPU box only (never the measured configuration): LRL_BENCH_REHEARSE=1 puts every rank on GPU 0 and
    # combines over gloo, to run the N > 1 code path end to end where RCCL cannot (it refuses two ranks on one device)
    rehearse = world > 1 and os.environ.get("LRL_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(180, exit=True)     # a hang names its line
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    # ---- scene: built on the host by the restated reference library, uploaded once ----
    t0 = time.time()
    c1, c3 = a.workload == "c1", a.workload == "c3"
    a.width = a.width or (1024 if c1 else 1920)
    a.height = a.height or (1024 if c1 else 1080)
    a.depth = a.depth or (10 if c1 else 8 if c3 else 4)
    a.spp = a.spp or (64 if c3 else 1)
    variant = abi.VARIANT_POOL if c3 else abi.VARIANT_BVH
    if c3:
        m = scenes.pool(builder="reference" if a.builder == "device" else a.builder)      # (the Pool model always builds its tree; not handed over)
    elif c1:
        m = scenes.bvh_default(builder="reference" if a.builder == "device" else a.builder)
    else:
        m = scenes.mesh_scene(subdivisions=a.subdivisions, builder="none" if a.builder == "device" else a.builder, grid=a.grid)
    build_s = time.time() - t0
    info = m.bvh_info() if a.builder != "device" or c3 or c1 else {"nodes": 0, "max_depth": 0}
    W, H = a.width, a.height
    trans = m.camera(H)
    view = m.scene_view()
    if a.builder == "device":
        view.bvh = None
        view.bvhNodeNum = 0
    tr = tracer.Tracer(local_rank)
    t0 = time.time()
    tr.scene_upload(view)
    upload_s = time.time() - t0
    tr.set_camera(trans, W, H)
    tr.set_decay_origin((0, 0, 0, 1))
    mode = abi.MODE_FAST if a.mode == "fast" else abi.MODE_PARITY
    # every rank renders a.spp samples of an (a.spp * world)-entry jitter table: per-GPU work is fixed (weak scaling)
    total_spp = a.spp * world
    jitter = host.jitter_table(total_spp) if total_spp > 1 else None
    samples = (rank * a.spp, (rank + 1) * a.spp) if world > 1 else None

    # ---- ray / node / primitive-test counts of this rank's frame (counting variant, un-timed) ----
    _, _ = tr.render(max_depth=a.depth, variant=variant, mode=mode, counters=True, jitter=jitter, samples=samples, want_hits=False, diffuse=a.diffuse)
    cn = tr.counters()
    rays, alg_bytes = cn.rays, cn.algorithmic_bytes()

    out = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
    # N > 1: two image buffers, so that the all-reduce of frame i (RCCL's own stream) runs while frame i + 1 renders.  The
    # render stream waits for the collective that last used a buffer before rendering into it again (stream-level waits
    # only, no host wait inside the loop); every step still does one render and one all-reduce.
    outs = [out, torch.empty_like(out)] if world > 1 else [out]
    works = [None, None]
    render_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    counter = [0]

    # LRL_BENCH_OVERLAP=1: the all-reduce of frame i overlaps the render of frame i + 1 (two buffers, one collective in
    # flight).  Not the default: its two-rank rehearsal over gloo (whose collectives block a worker thread inside HIP waits)
    # stalled now and then in the timed loop, and a stalled scaling run measures nothing; RCCL has no such thread, so the
    # overlap is there to be switched on once it has run on a multi-GPU node.
    overlap = os.environ.get("LRL_BENCH_OVERLAP") == "1"

    def step():
        if world == 1:
            tr.render_device(out.data_ptr(), max_depth=a.depth, variant=variant, mode=mode, jitter=jitter, samples=samples, diffuse=a.diffuse)
            return
        if not overlap:
            # default: the frame and its all-reduce one after the other on one stream (RCCL: both are only enqueued)
            counter[0] += 1
            with torch.cuda.stream(render_stream):
                tr.render_device(outs[0].data_ptr(), stream=render_stream.cuda_stream, max_depth=a.depth, variant=variant, mode=mode,
                                 jitter=jitter, samples=samples, diffuse=a.diffuse)
                dist.all_reduce(outs[0], op=dist.ReduceOp.SUM)
            if rehearse:
                print(f"[rehearsal] rank {rank}: step {counter[0]} done", file=sys.stderr, flush=True)
            return
        b = counter[0] & 1
        counter[0] += 1
        with torch.cuda.stream(render_stream):
            # frame i renders (into the buffer whose all-reduce, of frame i - 2, was waited for in the last step) while the
            # all-reduce of frame i - 1 runs; then that one is waited for (RCCL: a wait of the render stream, not of the
            # host) and the all-reduce of frame i is queued behind its render: never more than one collective in flight
            tr.render_device(outs[b].data_ptr(), stream=render_stream.cuda_stream, max_depth=a.depth, variant=variant, mode=mode,
                             jitter=jitter, samples=samples, diffuse=a.diffuse)
            if works[1 - b] is not None:
                works[1 - b].wait()
                works[1 - b] = None
            works[b] = dist.all_reduce(outs[b], op=dist.ReduceOp.SUM, async_op=True)
        if rehearse:
            print(f"[rehearsal] rank {rank}: step {counter[0]} queued", file=sys.stderr, flush=True)
            if os.environ.get("LRL_BENCH_REHEARSE_SYNC") == "1":      # tells a render that never ends from a collective that never ends
                render_stream.synchronize()
                print(f"[rehearsal] rank {rank}: step {counter[0]} rendered", file=sys.stderr, flush=True)

    def drain():
        for i in (0, 1):
            if works[i] is not None:
                works[i].wait()
                works[i] = None
        tr.sync()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tr.timing_begin()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    frame_ms_total, frames = tr.timing_end()

    # ---- trace-kernel time of one frame (HIP events around every trace launch), separate short pass ----
    # (the library keeps at most 4096 timed launches: one frame is enough where a frame has hundreds of them)
    TRACE_FRAMES = 5 if a.spp <= 64 else 1
    tr.timing_begin()
    for _ in range(TRACE_FRAMES):
        tr.render_device(out.data_ptr(), max_depth=a.depth, variant=variant, mode=mode, jitter=jitter, samples=samples, time_kernels=True, diffuse=a.diffuse)
    tr.sync()
    trace_ms_total, trace_launches = tr.timing_end()

    tot = torch.tensor([elapsed, float(rays), float(alg_bytes)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        rays_all = float(tot[1])
    else:
        rays_all = float(rays)

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = rays_all * a.steps / elapsed / 1e6
        frame_ms = frame_ms_total / max(frames, 1)
        kernel_ms = trace_ms_total / TRACE_FRAMES
        if trace_launches >= 4096:
            raise SystemExit("bench.py: more trace launches per timed pass than the library keeps events for; lower --spp")
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r01_traffic.json")        # PMC passes are separate runs (tools/run_pmc.sh)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if (tj.get("workload_triangles") == int(view.num.triangleNum) and tj.get("mode") == a.mode and tj.get("builder") == a.builder
                    and not a.diffuse and (W, H, a.spp, a.depth) == (1920, 1080, 1, 4)):      # what the PMC passes ran
                traffic = tj.get("hbm_bytes_per_frame")
        dbytes = tr.scene_device_bytes()
        line = {
            "metric": "Mrays/s (whole node) at 1080p 4-bounce", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"C1: the reference's default BVH scene ({view.num.triangleNum} triangles + spheres, discs, cylinders, cones), " if c1 else
                                    f"C3: Pool demo scene ({view.num.triangleNum} triangles, water at rest), Pool shader variant, " if c3 else
                                    f"C2: Stanford bunny x{a.subdivisions} subdivisions{f', {a.grid}x{a.grid} copies' if a.grid > 1 else ''} ({view.num.triangleNum} triangles) + ground disc + 1 light, ") +
                                   f"{W}x{H}, {a.spp} spp per GPU, RayTraceDepth {a.depth}" + (", diffuse bounces (Philox, cosine-weighted)" if a.diffuse else ""),
                       "bvh_builder": a.builder, "bvh_nodes": info["nodes"], "bvh_depth": info["max_depth"], "arithmetic": a.mode,
                       "parallelism": f"replicated scene, {a.spp} sample indices per GPU, RCCL all-reduce of the image" if world > 1 else "1 GPU",
                       "rays_per_frame": int(rays), "host_scene_build_s": round(build_s, 2), "scene_upload_s": round(upload_s, 2),
                       "device_scene_bytes": dbytes},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "kernel_ms": round(kernel_ms, 4),
                         "kernel": "trace kernels of one frame (wfTraceLeanKernel + wfTraceGroupKernel)" if a.mode == "fast" else "trace kernels of one frame (wfTraceKernel)",
                         "launches_per_frame": trace_launches // TRACE_FRAMES, "frame_ms": round(frame_ms, 4),
                         "frame_frac": round(alg_bytes / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                         "algorithmic_bytes_per_frame": int(alg_bytes),
                         "counts": {"node_visits": int(cn.nodeVisits), "tri_tests": int(cn.triTests), "sphere_tests": int(cn.sphereTests),
                                    "other_tests": int(cn.otherTests), "rays": int(rays)}},
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(a, m, view, trans, W, H, variant)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    tr.close()


def cpu_baseline(a, m, view, trans, W, H, variant):
    """The CPU oracle (C++ restatement of the reference shader, OpenMP over rows) on a bounded stripe of the SAME frame."""
    from oracle import oracle_api as O
    cores = min(16, os.cpu_count() or 1)
    rows = a.cpu_rows or H
    r0 = max(0, (H - rows) // 2)
    # repeat the stripe until ~12 s of CPU work have been timed (at least twice)
    rays, dt, reps = 0, 0.0, 0
    while dt < 12.0 or reps < 2:
        t0 = time.perf_counter()
        _, _, c = O.render(view, trans, W, H, max_depth=a.depth, variant=variant, rows=(r0, r0 + rows), threads=cores, want_hits=False, diffuse=a.diffuse)
        dt += time.perf_counter() - t0
        rays += c.rays
        reps += 1
        if reps >= 200:
            break
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"rows {r0}..{r0 + rows} of the same {W}x{H} frame (one sample per pixel) rendered {reps} times ({rays} rays, {dt:.1f} s), "
                      f"oracle/liboracle.so, OpenMP over rows"}


if __name__ == "__main__":
    main()
