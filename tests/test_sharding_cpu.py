"""CPU tests of the multi-GPU path: batch x head sharding and the verification gather.

The attention kernel itself needs a GPU; what N > 1 adds is (a) which heads each
rank owns and (b) the all-gather of per-head checksums used to verify a sharded
run against an unsharded one.  Both are exercised here with world_size 2 on the
gloo backend, with a plain torch CPU attention standing in for the kernel (it is
the test's stand-in for the compute, not a product path).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from my_flash_attn_cuda import sharding


def test_head_range_partitions_exactly():
    for total in (1, 7, 8, 128, 256, 257):
        for world in (1, 2, 3, 4, 8):
            spans = [sharding.head_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.head_range(8, 2, 2)


def test_shard_bhnd_is_contiguous_slice():
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    parts = [sharding.shard_bhnd(x, r, 4) for r in range(4)]
    assert all(p.dim() == 4 and p.shape[0] == 1 and p.is_contiguous() for p in parts)
    assert torch.equal(torch.cat([p[0] for p in parts], dim=0), x.reshape(6, 4, 5))


def _cpu_attention(q, k, v, scale):
    s = torch.matmul(q.double(), k.double().transpose(-1, -2)) * scale
    return torch.matmul(torch.softmax(s, dim=-1), v.double()).float()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, H, N, d, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)                      # every rank builds the same full problem ...
        q, k, v = (torch.randn(B, H, N, d) for _ in range(3))
        qs, ks, vs = (sharding.shard_bhnd(t, rank, world) for t in (q, k, v))   # ... and keeps its heads
        o = _cpu_attention(qs, ks, vs, d ** -0.5)
        sums = sharding.gather_head_checksums(o, B * H, rank, world)
        out_q.put((rank, sums))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B,H", [(2, 3), (1, 5)])
def test_two_rank_gather_matches_unsharded(B, H):
    world, N, d = 2, 48, 16
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, H, N, d, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out_q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d) for _ in range(3))
    full = _cpu_attention(q, k, v, d ** -0.5).reshape(1, B * H, N, d)
    want = sharding.head_checksums(full)
    for r in range(world):
        assert got[r].shape == (B * H, 2)
        assert torch.allclose(got[r], want, rtol=1e-12, atol=1e-12)
    assert torch.equal(got[0], got[1])
