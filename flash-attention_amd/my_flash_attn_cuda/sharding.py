"""Batch x head sharding of the attention forward across the GPUs of one node.

Every (batch, head) is an independent attention problem, so the flattened B*H
axis is split into contiguous ranges, one per rank (one process per GPU), and
the hot path needs no collective at all.  The only communication is for
verification: an all-gather of per-head checksums (or of the outputs) through
torch.distributed -- backend "nccl", which is RCCL over xGMI on ROCm, or "gloo"
in the CPU tests.  The reference has no multi-GPU code; this is new surface.
"""
import torch
import torch.distributed as dist


def head_range(total_heads, rank, world):
    """Contiguous [start, stop) slice of the flattened B*H axis owned by `rank`.

    Ranges differ in length by at most one head and cover the axis exactly once.
    """
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(int(total_heads), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_bhnd(x, rank, world):
    """The rank's shard of a (B, H, N, d) tensor, as a (1, heads_here, N, d) view/copy."""
    B, H, N, d = x.shape
    s, e = head_range(B * H, rank, world)
    return x.reshape(B * H, N, d)[s:e].unsqueeze(0).contiguous()


def head_checksums(o):
    """Per-head float64 checksum vector [sum, sum of squares] of a (1, heads, N, d) output."""
    f = o.double().flatten(2)
    return torch.stack([f.sum(dim=2)[0], (f * f).sum(dim=2)[0]], dim=1)


def gather_head_checksums(o, total_heads, rank, world, group=None, force=False):
    """All-gathers the per-head checksums of every rank's shard into one (total_heads, 2)
    tensor in head order (equal on all ranks).  Verification only; not on the timed path.
    `force` runs the collective even in a world of one (rehearsal)."""
    mine = head_checksums(o)
    full = torch.zeros((total_heads, 2), dtype=torch.float64, device=o.device)
    s, e = head_range(total_heads, rank, world)
    full[s:e] = mine
    if world > 1 or force:
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)   # disjoint rows: sum == gather
    return full
