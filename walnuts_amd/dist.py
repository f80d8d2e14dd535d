"""Sharding chains over the GPUs of one node: contiguous blocks, one process per
GPU, no exchange during sampling; the only collective is the final gather of draws (RCCL when the
backend is "nccl"; the same code runs on "gloo" for the CPU tests)."""


def shard_range(n_chains_total, rank, world):
    """Contiguous block of global chain ids owned by `rank`: [lo, hi)."""
    base, rem = divmod(int(n_chains_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_draws(local, dst=0):
    """Gather per-rank draw tensors [C_r, ...] on `dst` and concatenate along the chain axis.
    Ranks may hold different numbers of chains.  Returns the concatenation on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cmax = max(sizes)
    pad = local
    if local.shape[0] < cmax:
        pad = torch.cat([local, local.new_zeros((cmax - local.shape[0],) + tuple(local.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


def total_steps(diag):
    """Leapfrog steps (gradient evaluations) summed over every chain of every rank:
    sum(diagnostics[:,6] + diagnostics[:,7]) (WALNUTS.py:676-677)."""
    import torch.distributed as dist
    t = (diag[..., 6].sum() + diag[..., 7].sum()).reshape(1).clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
