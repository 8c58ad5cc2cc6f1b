"""Multi-GPU sharding of one frame.  The scene is replicated on every rank; pixels and
samples are independent, so the only exchange step is combining the per-rank image parts:

  * row stripes   — rank r renders interleaved stripes of `stripe` rows; parts are all-gathered and
                    interleaved back.  Bit-identical to the single-GPU frame.
  * interleaved   — the same stripes rendered by ONE call per rank (LrlRenderParams.rowInterleave / rowShard: the rank's
                    stripes come back concatenated), all-gathered with one collective and un-interleaved with one
                    index_select.  Bit-identical to the single-GPU frame; this is the strong-scaling path of bench.py.
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


STRIPE_ROWS = 16      # LRL_STRIPE_ROWS (include/lrl_trace.h)


def shard_rows(height, world, rank, stripe=STRIPE_ROWS):
    """Rows of rank's interleaved shard (== lrl_shard_rows)."""
    return sum(r1 - r0 for r0, r1 in stripes(height, world, rank, stripe))


def interleave_index(height, world, stripe=STRIPE_ROWS):
    """For every row of the frame, its row in the all-gathered buffer [world * pad, W, 4] (pad = the largest shard)."""
    pad = max(shard_rows(height, world, k, stripe) for k in range(world))
    idx = torch.empty(height, dtype=torch.long)
    for k in range(world):
        o = 0
        for r0, r1 in stripes(height, world, k, stripe):
            idx[r0:r1] = torch.arange(k * pad + o, k * pad + o + (r1 - r0))
            o += r1 - r0
    return idx, pad


def render_interleaved(render, height, width, world, rank, device="cpu", index=None):
    """Every rank returns the full [H, W, 4] frame.  render(rows=("interleave", world, rank)) -> this rank's stripes, concatenated,
    as a tensor [shard_rows, W, 4] on `device` (or writes into the buffer handed to it: render(rows=..., out=buf))."""
    if index is None:
        index = interleave_index(height, world)
    idx, pad = index
    mine = shard_rows(height, world, rank)
    buf = torch.zeros((pad, width, 4), dtype=torch.float32, device=device)
    if mine:
        buf[:mine] = render(rows=("interleave", world, rank))
    if world > 1:
        full = torch.empty((world * pad, width, 4), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(full, buf)
    else:
        full = buf
    return full.index_select(0, idx.to(full.device))
