"""Multi-GPU sharding of one frame.  The scene is replicated on every rank; pixels and
samples are independent, so the only exchange step is combining the per-rank image parts:

  * row stripes   — rank r renders interleaved stripes of `stripe` rows; parts are all-gathered and
                    interleaved back.  Bit-identical to the single-GPU frame.
  * sample ranges — rank r renders samples [s0, s1) of the jitter table for the whole frame; each part
                    is already divided by the total spp, so an all-reduce(SUM) gives the mean.

`render(rows=(r0, r1))` / `render(samples=(s0, s1))` is any callable returning a torch tensor
[rows, W, 4] on the device the process group works on (the HIP tracer on GPUs with the nccl = RCCL
backend; the CPU oracle under gloo in tests/test_sharding_gloo.py).
"""
import torch
import torch.distributed as dist


def stripes(height, world, rank, stripe=16):
    """Interleaved row stripes of this rank: [(r0, r1), ...]; every row belongs to exactly one rank."""
    out = []
    for i, r0 in enumerate(range(0, height, stripe)):
        if i % world == rank:
            out.append((r0, min(height, r0 + stripe)))
    return out


def sample_range(spp, world, rank):
    """Contiguous, balanced split of sample indices; ranks beyond spp get an empty range."""
    base, extra = divmod(spp, world)
    s0 = rank * base + min(rank, extra)
    return s0, s0 + base + (1 if rank < extra else 0)


def render_rows(render, height, width, world, rank, stripe=16, device="cpu"):
    """Every rank returns the full [H, W, 4] frame."""
    mine = stripes(height, world, rank, stripe)
    parts = [render(rows=r) for r in mine]
    local = torch.cat(parts, dim=0) if parts else torch.zeros((0, width, 4), dtype=torch.float32, device=device)
    # all ranks contribute equally sized buffers: pad to the largest share
    counts = [sum(r1 - r0 for r0, r1 in stripes(height, world, k, stripe)) for k in range(world)]
    pad = max(counts)
    buf = torch.zeros((pad, width, 4), dtype=torch.float32, device=device)
    buf[:local.shape[0]] = local
    if world > 1:
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
    else:
        gathered = [buf]
    frame = torch.empty((height, width, 4), dtype=torch.float32, device=device)
    for k in range(world):
        o = 0
        for r0, r1 in stripes(height, world, k, stripe):
            frame[r0:r1] = gathered[k][o:o + (r1 - r0)]
            o += r1 - r0
    return frame


def render_samples(render, spp, height, width, world, rank, device="cpu"):
    """Every rank returns the mean image over all spp samples."""
    s0, s1 = sample_range(spp, world, rank)
    part = render(samples=(s0, s1)) if s1 > s0 else torch.zeros((height, width, 4), dtype=torch.float32, device=device)
    if world > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM)
    return part
