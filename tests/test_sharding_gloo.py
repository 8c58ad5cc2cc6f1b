"""CPU, world_size 2, gloo: the multi-GPU combine logic of raytracing_amd/sharding.py with the CPU oracle
standing in for the tracer (the GPU path uses the same functions with the nccl backend in bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import sharding


def test_stripes_partition_rows():
    for H in (1, 15, 16, 17, 135, 1080):
        for world in (1, 2, 3, 8):
            rows = sorted(r for k in range(world) for a, b in sharding.stripes(H, world, k) for r in range(a, b))
            assert rows == list(range(H))


def test_sample_ranges_partition_samples():
    for spp in (1, 2, 7, 64):
        for world in (1, 2, 8):
            got = [s for k in range(world) for s in range(*sharding.sample_range(spp, world, k))]
            assert got == list(range(spp))


def test_interleave_index_matches_the_stripes():
    for H in (1, 16, 17, 135, 1080, 2160):
        for world in (1, 2, 3, 8):
            idx, pad = sharding.interleave_index(H, world)
            assert pad == max(sharding.shard_rows(H, world, k) for k in range(world))
            assert sum(sharding.shard_rows(H, world, k) for k in range(world)) == H
            # gathering shard k's rows back through the index gives the rows the stripes name
            rows = torch.full((world * pad,), -1, dtype=torch.long)
            for k in range(world):
                mine = [r for a, b in sharding.stripes(H, world, k) for r in range(a, b)]
                rows[k * pad:k * pad + len(mine)] = torch.tensor(mine, dtype=torch.long)
            assert torch.equal(rows[idx], torch.arange(H))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_api as O
    from raytracing_amd import host, scenes
    m = scenes.bvh_default()
    scenes.overview_camera(m)
    W, H, spp = 80, 56, 4
    tr = m.camera(H)
    view = m.scene_view()
    jit = host.jitter_table(spp)
    ts = host.sample_trans(tr, jit)

    def render_rows(rows):
        img, _, _ = O.render(view, tr, W, H, max_depth=6, rows=rows, want_hits=False)
        return torch.from_numpy(img)

    def render_samples(samples):
        s0, s1 = samples
        acc = np.zeros((H, W, 4), dtype=np.float32)
        for s in range(s0, s1):                      # partial sum / spp, as LrlRenderParams.sampleBegin/End defines it
            img, _, _ = O.render(view, ts[s:s + 1], W, H, max_depth=6, want_hits=False)
            acc += img
        return torch.from_numpy(acc / np.float32(spp))

    frame = sharding.render_rows(lambda rows: render_rows(rows), H, W, world, rank, stripe=16)

    def render_shard(rows):                          # ("interleave", N, k): the oracle renders stripe by stripe
        _, n, k = rows
        return torch.cat([render_rows(r) for r in sharding.stripes(H, n, k)], dim=0)

    frame2 = sharding.render_interleaved(render_shard, H, W, world, rank)
    mean = sharding.render_samples(lambda samples: render_samples(samples), spp, H, W, world, rank)
    if rank == 0:
        full, _, _ = O.render(view, tr, W, H, max_depth=6, want_hits=False)
        full_mean, _, _ = O.render(view, ts, W, H, max_depth=6, want_hits=False)
        q.put((bool(np.array_equal(frame.numpy(), full)) and bool(np.array_equal(frame2.numpy(), full)), float(np.abs(mean.numpy()[..., :3] - full_mean[..., :3]).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_compose_the_single_process_frame():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    stripes_equal, mean_err = q.get()
    assert stripes_equal                      # stripes: bit-identical
    assert mean_err <= 2e-6                   # sample partial means: float add order only
