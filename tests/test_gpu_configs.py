"""GPU tests at the shapes of BASELINE.json's larger configurations (C3, C4, C5), through the C ABI, and of the multi-GPU
sharding units driven by the HIP tracer itself.  Whole-frame oracle runs are out of reach at these sizes, so each test combines
size-independent properties (FAST against PARITY, idempotence, stripes / shards / sample ranges composing the frame, ray counts)
with an oracle run on a stripe of the same frame.  The path under test: /root/reference/BVH/shaders/RendererFragment0.cpp:406-632
(traversal) and :633-740 (shading loop), Pool/Pool.cpp:637-672 (accumulation over frames)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ID_BAR, RMSE_PARITY, check_image, image_stats, load_golden
from oracle import oracle_api as O
from raytracing_amd import abi, host, scenes, sharding

pytestmark = pytest.mark.gpu



def without_tree(model):
    v = model.scene_view()
    v.bvh = None
    v.bvhNodeNum = 0
    return v


def tie_adjusted_match(a, b, view, trans, W, H, row0=0, pixels=None):
    """oracle_api.tie_adjusted_match: share of pixels whose primary hits agree or are a proven floating-point tie (both primitives met
    at the same distance within the float32 rounding band, or the nearer one met at its boundary).  Far from a scene of sub-pixel
    triangles the rounding band of an edge, eps x distance, is a visible fraction of a triangle, so FAST (rcp, fma) and PARITY (IEEE
    divide) hand a good many edge pixels to different neighbours.  -> (share, exact share, number of pixels that are neither)"""
    share, exact, bad = O.tie_adjusted_match(a, b, view, trans, W, H, pixels=pixels, row0=row0)
    return share, exact, int(len(bad))


def test_c5_regime(tracer):
    """C5's regime at a size a test can afford: 5 x 5 replicated copies of the subdivided bunny (31 M triangles, 2 GB of node
    records: beyond every cache), tree built on the device, diffuse (incoherent) secondary rays, 1920 x 1080, 2 spp, depth 4.  At
    this size the FAST pipeline switches on, by itself, the compaction of the late segments, the later hand-over (96 / 48 trips) and
    512-entry frontiers for the 16-lanes-per-ray hand-over kernel (wfTraceGroupKernel<16, 2>); all asserted, not assumed, and the
    wave-per-ray kernel with its 1024-entry frontier (and its rounds of four lanes to an entry) is run on the same frame."""
    m = scenes.mesh_scene(subdivisions=2, builder="none", grid=5)
    W, H = 1920, 1080
    tr = m.camera(H)
    tracer.scene_upload(without_tree(m))
    info = tracer.scene_info()
    assert info["deviceBuilt"] == 1 and info["bvhRecords"] * 64 > (1 << 30), info
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    jit = host.jitter_table(2)
    par, phits = tracer.render(max_depth=4, mode=abi.MODE_PARITY, jitter=jit, diffuse=True, counters=True)
    rays = tracer.counters().rays
    assert tracer.counters().raysPrimary == 2 * W * H and rays > 4 * W * H
    fast, fhits = tracer.render(max_depth=4, mode=abi.MODE_FAST, jitter=jit, diffuse=True)
    ri = tracer.last_render_info()
    assert ri["compactFrom"] == 2 and ri["groupLanes"] == 16 and ri["frontierEntries"] == 512 and ri["batch"] == 2, ri
    assert ri["handoverTrips"][0] >= 96 and ri["handoverTrips"][3] >= 48, ri
    # the wave-per-ray kernel with its 1024-entry frontier, the lanes-per-ray kernel with four lanes to an entry, and the early
    # hand-over of scenes in the caches: the same frame bit for bit
    base = tracer.get_tuning()
    try:
        for kw, lanes, entries in (({"groupLanes": 64}, 64, 1024), ({"groupQuad": 1}, 16, 512), ({"groupLanes": 32, "groupCapX": 2, "handoverEarly": 48, "handoverLate": 16}, 32, 512)):
            tracer.set_tuning(base)
            tracer.set_tuning(**kw)
            fast2, fhits2 = tracer.render(max_depth=4, mode=abi.MODE_FAST, jitter=jit, diffuse=True)
            ri2 = tracer.last_render_info()
            assert ri2["groupLanes"] == lanes and ri2["frontierEntries"] == entries, ri2
            assert np.array_equal(fast2.view(np.uint32), fast.view(np.uint32)) and np.array_equal(fhits2, fhits), kw
    finally:
        tracer.set_tuning(base)
    view = m.scene_view()
    ts0 = host.sample_trans(tr, jit[:1])               # the ids are sample 0's
    same, exact, others = tie_adjusted_match(fhits, phits, view, ts0, W, H)
    st = image_stats(fast, par)
    # observed: 99.983 % identical, the rest pairs of triangles sharing an edge (the camera is 5 scene widths away)
    assert same >= ID_BAR and exact >= 0.999, (same, exact, others)
    # diffuse bounces: a direction that moves by an ulp may send a path elsewhere, so FAST is judged by the share of pixels that moved
    assert st["frac_gt_1e-3"] <= 5e-3 and np.isfinite(fast).all(), st
    # idempotence and stripe invariance (PARITY: bit for bit)
    again, _ = tracer.render(max_depth=4, mode=abi.MODE_PARITY, jitter=jit, diffuse=True)
    assert np.array_equal(again, par)
    r0 = 528
    stripe, shits = tracer.render(max_depth=4, mode=abi.MODE_PARITY, jitter=jit, diffuse=True, rows=(r0, r0 + 8))
    assert np.array_equal(stripe, par[r0:r0 + 8]) and np.array_equal(shits, phits[r0:r0 + 8])
    fstripe, _ = tracer.render(max_depth=4, mode=abi.MODE_FAST, jitter=jit, diffuse=True, rows=(r0, r0 + 8))
    assert image_stats(fstripe, fast[r0:r0 + 8])["frac_gt_1e-3"] <= 5e-3
    # the oracle on that stripe, over a host-built median tree (the image does not depend on the tree)
    m.data_init("median")
    ts = host.sample_trans(tr, jit)
    oimg, ohits, oc = O.render(m.scene_view(), ts, W, H, max_depth=4, rows=(r0, r0 + 8), threads=16, diffuse=True)
    osame, oexact, oothers = tie_adjusted_match(shits, ohits, m.scene_view(), ts0, W, H, row0=r0)
    assert osame >= ID_BAR and oexact >= 0.999, (osame, oexact, oothers)
    ost = image_stats(stripe, oimg)
    assert ost["frac_gt_1e-3"] <= 2e-3, ost


def test_c5_at_its_own_size(tracer):
    """C5's scene itself: 9 x 9 copies of the subdivided bunny, 100.5 M triangles, tree built on the device (12.9 GB of node records, 6.4 GB
    of hot triangle records: byte offsets pass 2^32 three times over, which no smaller scene does), 1920 x 1080, diffuse bounces, depth 4.
    Checked three ways: (i) a TREE-FREE brute-force oracle (every one of the 100.5 M triangles against the primary rays of 256 scattered
    pixels, oracle_bruteforce_primary: no tree, no node index, no offset to get wrong) against the PARITY and FAST primary ids at those
    pixels, with the hit distances; (ii) FAST against PARITY ids over the whole frame, mismatches only at shared edges / vertices;
    (iii) idempotence and a stripe.  Path: /root/reference/BVH/shaders/RendererFragment0.cpp:406-632."""
    m = scenes.mesh_scene(subdivisions=2, builder="none", grid=9)
    W, H = 1920, 1080
    tr = m.camera(H)
    view = m.scene_view()
    assert view.num.triangleNum == 81 * 1240960
    tracer.scene_upload(without_tree(m))
    info = tracer.scene_info()
    assert info["deviceBuilt"] == 1 and info["bvhRecords"] * 64 > (1 << 32), info
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    par, phits = tracer.render(max_depth=4, mode=abi.MODE_PARITY, diffuse=True)
    fast, fhits = tracer.render(max_depth=4, mode=abi.MODE_FAST, diffuse=True)
    assert tracer.last_render_info()["compactFrom"] == 2
    # (i) the brute-force oracle
    px = O.scattered_pixels(W, H, 256)
    bhits, bt = O.render_bruteforce(view, tr, W, H, px, threads=16)
    on_mesh = int((bhits[:, 0] == 2).sum())
    assert on_mesh >= 32, on_mesh                      # the sample looks at the mesh, not only at the ground and the sky
    for name, hits in (("PARITY", phits), ("FAST", fhits)):
        same, exact, others = tie_adjusted_match(hits[px[:, 1], px[:, 0]], bhits, view, tr, W, H, pixels=px)
        assert others == 0 and exact >= 0.98, (name, same, exact, others)
    # (ii) FAST against PARITY over the frame
    same, exact, others = tie_adjusted_match(fhits, phits, view, tr, W, H)
    assert same >= ID_BAR and exact >= 0.999, (same, exact, others)
    st = image_stats(fast, par)
    assert st["frac_gt_1e-3"] <= 5e-3 and np.isfinite(fast).all(), st
    # (iii) idempotence; a stripe in the middle of the frame
    again, ahits = tracer.render(max_depth=4, mode=abi.MODE_PARITY, diffuse=True)
    assert np.array_equal(again, par) and np.array_equal(ahits, phits)
    fagain, fahits = tracer.render(max_depth=4, mode=abi.MODE_FAST, diffuse=True)
    assert np.array_equal(fagain.view(np.uint32), fast.view(np.uint32)) and np.array_equal(fahits, fhits)
    r0 = 536
    stripe, shits = tracer.render(max_depth=4, mode=abi.MODE_PARITY, diffuse=True, rows=(r0, r0 + 8))
    assert np.array_equal(stripe, par[r0:r0 + 8]) and np.array_equal(shits, phits[r0:r0 + 8])


def test_c4_shape(tracer):
    """C4's shape: the C2 scene at 3840 x 2160 with samples of a 256-entry Halton table.  Four of its samples: sample ranges
    (0,2) + (2,4) add up to the range (0,4); three row bands and eight interleaved shards of 16-row stripes each compose the
    frame bit for bit (PARITY); FAST agrees with PARITY; the oracle agrees on a stripe."""
    m = scenes.mesh_scene(subdivisions=2, builder="reference")
    W, H = 3840, 2160
    tr = m.camera(H)
    view = m.scene_view()
    tracer.scene_upload(view)
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    jit = host.jitter_table(256)
    kw = dict(max_depth=4, jitter=jit, want_hits=False)
    full, hits = tracer.render(max_depth=4, jitter=jit, samples=(0, 4))
    assert np.all(full[..., 3] == np.float32(4 / 256))              # alpha = the share of the samples summed
    a, _ = tracer.render(samples=(0, 2), **kw)
    b, _ = tracer.render(samples=(2, 4), **kw)
    np.testing.assert_allclose(a + b, full, rtol=0, atol=1e-6)
    bands = [tracer.render(samples=(0, 4), rows=r, **kw)[0] for r in ((0, 700), (700, 1400), (1400, 2160))]
    assert np.array_equal(np.concatenate(bands), full)
    N = 8
    idx, pad = sharding.interleave_index(H, N)
    gathered = np.zeros((N * pad, W, 4), dtype=np.float32)
    for k in range(N):
        part, _ = tracer.render(samples=(0, 4), rows=("interleave", N, k), **kw)
        assert part.shape[0] == sharding.shard_rows(H, N, k) == tracer.shard_rows(N, k)
        gathered[k * pad:k * pad + part.shape[0]] = part
    assert np.array_equal(gathered[idx.numpy()], full)
    fast, fhits = tracer.render(max_depth=4, jitter=jit, samples=(0, 4), mode=abi.MODE_FAST)
    same = (fhits == hits).all(axis=-1).mean()
    assert same >= ID_BAR, same
    check_image(fast * 64, full * 64, abi.MODE_FAST, "C4 shape, FAST vs PARITY")
    r0 = 1000
    ts = host.sample_trans(tr, jit[:4])
    oimg, ohits, _ = O.render(view, ts, W, H, max_depth=4, rows=(r0, r0 + 8), threads=16)
    assert (hits[r0:r0 + 8] == ohits).all(axis=-1).mean() >= ID_BAR
    ost = image_stats(full[r0:r0 + 8] * 64, oimg)                   # the oracle's image is the mean of the four samples
    assert ost["rmse"] <= RMSE_PARITY, ost


@pytest.mark.parametrize("mode", [abi.MODE_PARITY, abi.MODE_FAST])
def test_c3_pool_halton_accumulation_vs_oracle(tracer, mode):
    """C3's shape: the Pool scene (Pool shader variant, RayTraceDepth 8: refraction stacks, textures) accumulated over 8 Halton
    samples, against the oracle's mean of the same 8 jittered frames (the reference accumulates nothing itself; its frame loop is
    Pool/Pool.cpp:637-672, the jitter is SURVEY 8d's shear of `trans`)."""
    pool_model = scenes.pool()                        # a model of its own: the session's fixtures keep their cameras
    W, H = 480, 270
    pool_model.camera_look_at(60.0, (0.42, -0.44, 0.9), (0.0, 0.1, 0.35))      # view B of the goldens
    tr = pool_model.camera(H)
    view = pool_model.scene_view()
    tracer.scene_upload(view)
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    jit = host.jitter_table(8)
    ts = host.sample_trans(tr, jit)
    oimg, ohits, oc = O.render(view, ts, W, H, max_depth=8, variant=abi.VARIANT_POOL, threads=16)
    img, hits = tracer.render(max_depth=8, variant=abi.VARIANT_POOL, mode=mode, jitter=jit, counters=(mode == abi.MODE_PARITY))
    same = (hits == ohits).all(axis=-1).mean()
    assert same >= ID_BAR, (mode, same)
    check_image(img, oimg, mode, "C3 shape: Pool, 8 Halton samples")
    if mode == abi.MODE_PARITY:
        assert abs(tracer.counters().rays - oc.rays) <= 2e-4 * oc.rays
        assert tracer.last_render_info()["batch"] == 8


def test_c3_at_its_own_size(tracer):
    """C3 itself, as bench.py times it: the Pool scene, its own camera, 1920 x 1080, 64 Halton samples, RayTraceDepth 8 (the FAST pipeline
    takes the frame in passes of a few samples each; asserted).  FAST against PARITY over the whole frame (ids tie-adjusted, image), and both
    against the oracle on an 8-row stripe of that frame at all 64 samples.  Then the frame after lrl_water_step (Pool/Pool.cpp:648-672: the
    water's triangles move on the device, the caller's four-wide tree is refitted behind the step by lrl_refit_node4): the same checks
    against the oracle's own water run."""
    from conftest import assert_ids
    m = scenes.pool()
    W, H, spp = 1920, 1080, 64
    tr = m.camera(H)
    view = m.scene_view()
    gx, gy = scenes.POOL_GRID
    k, dt = scenes.POOL_WATER
    nw = 2 * (gx - 1) * (gy - 1)
    tris = m.arrays()["triangles"]
    w0 = scenes.pool_water_points(m)
    tracer.scene_upload(view)
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    assert tracer.scene_info()["ownBoxes"] > 2000
    jit = host.jitter_table(spp)
    ts = host.sample_trans(tr, jit)
    ts0 = host.sample_trans(tr, jit[:1])
    kw = dict(max_depth=8, variant=abi.VARIANT_POOL, jitter=jit)
    r0 = 600                                           # a stripe across the water, the walls and the steps
    tracer.water_init(gx, gy, k, dt, w0, 0, tris[:nw])
    for steps in (0, 40):
        if steps:
            tracer.water_step(steps)
            ot = O.water_sync(O.water_run(w0, gx, gy, k, dt, steps), gx, gy, tris)
            v = _view_with(m, ot)
        else:
            v = view
        fast, fhits = tracer.render(mode=abi.MODE_FAST, **kw)
        ri = tracer.last_render_info()
        assert ri["passes"] > 1 and ri["passes"] * ri["batch"] >= spp, ri
        par, phits = tracer.render(mode=abi.MODE_PARITY, **kw)
        assert_ids(fhits, phits, v, ts0, W, H, ("C3 at its own size, FAST vs PARITY", steps))
        check_image(fast, par, abi.MODE_FAST, ("C3 at its own size, FAST vs PARITY", steps))
        oimg, ohits, _ = O.render(v, ts, W, H, max_depth=8, variant=abi.VARIANT_POOL, rows=(r0, r0 + 8), threads=16)
        assert_ids(phits[r0:r0 + 8], ohits, v, ts0, W, H, ("C3 at its own size, PARITY stripe vs oracle", steps), row0=r0)
        assert_ids(fhits[r0:r0 + 8], ohits, v, ts0, W, H, ("C3 at its own size, FAST stripe vs oracle", steps), row0=r0)
        check_image(par[r0:r0 + 8], oimg, abi.MODE_PARITY, ("C3 at its own size, PARITY stripe vs oracle", steps))
        check_image(fast[r0:r0 + 8], oimg, abi.MODE_FAST, ("C3 at its own size, FAST stripe vs oracle", steps))
        if steps:
            assert not np.array_equal(fast[r0:r0 + 8], still)      # (the water is in the stripe: the step was seen)
        still = fast[r0:r0 + 8].copy()


def _view_with(model, triangles):
    """The model's scene view with another triangle array (same count, same BVH: what the Pool does every frame)."""
    v = model.scene_view()
    keep = np.ascontiguousarray(triangles, dtype=abi.TRIANGLE)
    _view_with.keep = keep
    v.trianglesOrigin = keep.ctypes.data
    return v


def test_interleaved_shards_compose_the_frame(tracer):
    """LrlRenderParams.rowInterleave: N shards of 16-row stripes, each rendered by one call, give the frame bit for bit, with
    primary-hit ids, for shard counts that divide the stripes evenly and that do not, and a frame height that is no multiple of 16;
    diffuse bounces (keyed by the pixel of the full frame) included; a shard beyond the stripes is refused."""
    from raytracing_amd.tracer import TraceError
    default_model = scenes.bvh_default()              # a model of its own: the session's fixtures keep their cameras
    scenes.overview_camera(default_model)
    W, H = 200, 150                                   # 10 stripes, the last one 6 rows
    tr = default_model.camera(H)
    tracer.scene_upload(default_model.scene_view())
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    for diffuse in (False, True):
        full, fh = tracer.render(max_depth=6, diffuse=diffuse)
        for N in (2, 3, 10):
            idx, pad = sharding.interleave_index(H, N)
            img = np.zeros((N * pad, W, 4), dtype=np.float32)
            ids = np.zeros((N * pad, W, 2), dtype=np.uint32)
            for k in range(N):
                part, ph = tracer.render(max_depth=6, diffuse=diffuse, rows=("interleave", N, k))
                assert part.shape[0] == sharding.shard_rows(H, N, k)
                img[k * pad:k * pad + part.shape[0]] = part
                ids[k * pad:k * pad + part.shape[0]] = ph
            assert np.array_equal(img[idx.numpy()], full) and np.array_equal(ids[idx.numpy()], fh), (diffuse, N)
    with pytest.raises(TraceError):
        tracer.render(max_depth=6, rows=("interleave", 11, 10))       # 10 stripes: shard 10 of 11 holds nothing
    with pytest.raises(TraceError):
        tracer.render(max_depth=6, rows=("interleave", 2, 0), megakernel=True)


# ---- two ranks, both on GPU 0, gloo between them: the HIP tracer behind raytracing_amd/sharding.py ----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _hip_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracing_amd import tracer as T
    m = scenes.bvh_default()
    scenes.overview_camera(m)
    W, H, spp = 160, 120, 6
    tr = m.camera(H)
    t = T.Tracer(0)
    t.scene_upload(m.scene_view())
    t.set_camera(tr, W, H)
    t.set_decay_origin((0, 0, 0, 1))
    jit = host.jitter_table(spp)

    # the collectives run on host tensors: gloo's device path blocks a worker thread inside HIP waits, and RCCL refuses two ranks
    # on one device; the tracer calls, the shard geometry and the composition are the ones the multi-GPU bench uses
    def rows_part(rows):
        return torch.from_numpy(t.render(max_depth=6, rows=rows, want_hits=False)[0])

    def samples_part(samples):
        return torch.from_numpy(t.render(max_depth=6, jitter=jit, samples=samples, want_hits=False)[0])

    striped = sharding.render_rows(rows_part, H, W, world, rank)
    inter = sharding.render_interleaved(rows_part, H, W, world, rank)
    mean = sharding.render_samples(samples_part, spp, H, W, world, rank)
    if rank == 0:
        full = t.render(max_depth=6, want_hits=False)[0]
        full_mean = t.render(max_depth=6, jitter=jit, want_hits=False)[0]
        q.put((bool(np.array_equal(striped.numpy(), full)), bool(np.array_equal(inter.numpy(), full)),
               float(np.abs(mean.numpy()[..., :3] - full_mean[..., :3]).max()), float(np.abs(mean.numpy()[..., 3] - 1.0).max())))
    dist.barrier()
    t.close()
    dist.destroy_process_group()


def test_two_ranks_drive_the_hip_tracer():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    striped_equal, interleaved_equal, mean_err, alpha_err = q.get()
    assert striped_equal and interleaved_equal          # stripes: bit-identical to the one-rank frame
    assert mean_err <= 2e-6 and alpha_err <= 1e-6       # sample shards: float add order only; partial alphas add up to 1


# ---- bench.py's own N > 1 loop (the code the driver's 8-GPU run executes first), rehearsed with two ranks on this one GPU ----
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_rank_loop_rehearsal(scaling):
    """`python bench.py --gpus 2 --rehearse`: bench.py starts its two ranks itself (torch.distributed.run as a child process), both on
    GPU 0 with gloo on host copies in place of RCCL (which refuses two ranks on one device), and runs the same step / drain / barrier /
    max-over-ranks loop as the measured configuration: weak = sample ranges + reduce onto rank 0, strong = interleaved stripes +
    all-gather + un-interleave.  The JSON line must say what ran: two ranks, the gloo backend, and a whole-job rate that is the two
    ranks' rays over the slower rank's time.  Frame being sharded: /root/reference/BVH/BVH.cpp:38-43,741-742."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "c1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-c5",
                          "--scaling", scaling], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--rehearse", "--workload", "c1", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--scaling", scaling], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert two.returncode == 0, two.stderr[-2000:]
    l1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    l2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["scaling"] == scaling and l2["steps"] == 2
    cfg = l2["config"]
    assert cfg["collective_backend"] == "gloo" and cfg["ranks_seen_by_backend"] == 2 and "gloo" in cfg["parallelism"] and "RCCL" not in cfg["parallelism"]
    assert len(cfg["ms_per_step_per_rank"]) == 2 and abs(max(cfg["ms_per_step_per_rank"]) - l2["ms_per_step"]) <= 0.05 * l2["ms_per_step"] + 0.5
    # rays: weak = every rank the whole frame for its own samples (twice the one-rank frame); strong = the frame's rays once, split over the ranks
    r1, r2 = l1["config"]["rays_per_frame"], cfg["rays_per_frame"]
    if scaling == "weak":
        assert 1.9 * r1 <= r2 <= 2.1 * r1, (r1, r2)
    else:
        assert 0.95 * r1 <= r2 <= 1.05 * r1, (r1, r2)
    # value = rays of all ranks over the slower rank's time
    assert abs(l2["value"] - r2 / (l2["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * l2["value"]
    # two processes share one GPU and stage their collectives through the host: no speed bar, only sanity against the one-rank rate
    assert 0.02 * l1["value"] <= l2["value"] <= 2.5 * l1["value"], (l1["value"], l2["value"])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_rccl_step_with_one_rank(scaling):
    """`python bench.py --one-rank-rccl`: the step of the MEASURED multi-GPU configuration with the collective library it is measured
    with, on the one GPU a test has: a process group of one rank on RCCL (backend "nccl"), renders on a side stream,
    `dist.reduce(async_op=True)` overlapping the next frame (weak) or `all_gather_into_tensor` + un-interleave (strong), the barriers, the
    device-side max / sum over ranks.  One rank moves no bytes between GPUs; what this pins is that every RCCL call of the loop is issued
    with tensors, streams and an order the library accepts, and that the line names the backend that ran.  The frame and the rate must be
    the plain one-GPU run's (same rays; time within a factor that allows for the extra launches)."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    base = [sys.executable, os.path.join(REPO, "bench.py"), "--workload", "c1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-c5", "--scaling", scaling]
    one = subprocess.run(base, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert one.returncode == 0, one.stderr[-2000:]
    rccl = subprocess.run(base + ["--one-rank-rccl"], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert rccl.returncode == 0, rccl.stderr[-3000:]
    l1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    l2 = json.loads([l for l in rccl.stdout.splitlines() if l.startswith("{")][-1])
    cfg = l2["config"]
    assert l2["n_gpus"] == 1 and cfg["collective_backend"] == "nccl" and cfg["ranks_seen_by_backend"] == 1 and "RCCL" in cfg["parallelism"], cfg
    assert ("overlapping the next frame" in cfg["parallelism"]) == (scaling == "weak")
    assert cfg["rays_per_frame"] == l1["config"]["rays_per_frame"]
    assert 0.5 * l1["value"] <= l2["value"] <= 1.5 * l1["value"], (l1["value"], l2["value"])

