"""CPU tests of the multi-GPU path: the batch x head partition and the world_size-2 bookkeeping
over torch.distributed (gloo here; the same calls run over RCCL on the GPU box)."""
import os
import socket
import sys

import numpy as np
import pytest

import __graft_entry__ as entry

fa = entry.load_package()
from flash_attention_cuda_c_amd import shard  # noqa: E402


@pytest.mark.parametrize("BH,N", [(128, 8), (2048, 8), (7, 3), (1, 4), (100, 6), (0, 2)])
def test_shards_tile_the_head_range(BH, N):
    ranges = [shard.shard_heads(BH, r, N) for r in range(N)]
    assert ranges[0][0] == 0 and ranges[-1][1] == BH
    for (a, b), (c, d) in zip(ranges, ranges[1:]):
        assert b == c and a <= b
    sizes = [b - a for a, b in ranges]
    assert max(sizes) - min(sizes) <= 1
    off, cnt = shard.slab(BH, N - 1, N, 64, 16)
    assert off == ranges[-1][0] * 64 * 16 and cnt == sizes[-1] * 64 * 16


def test_bad_shard_arguments():
    with pytest.raises(ValueError):
        shard.shard_heads(8, 2, 2)
    with pytest.raises(ValueError):
        shard.shard_heads(8, 0, 0)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, entry.ROOT)
    import oracle  # the oracle stands in for the GPU kernel in this CPU test only
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, H, S, d = 2, 3, 48, 32
    rng = np.random.default_rng(11)                    # every rank regenerates the same global data
    Q, K, V = (rng.standard_normal((B * H, S, d), dtype=np.float32) for _ in range(3))
    lo, hi = shard.shard_heads(B * H, rank, world)
    mine = oracle.attention(Q[None, lo:hi], K[None, lo:hi], V[None, lo:hi], causal=True)[0]
    total = shard.reduce_sum(float(mine.astype(np.float64).sum()))
    worst = shard.reduce_max(1.0 + rank)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, mine, total, worst))


def test_world_size_2_gloo_sharded_equals_unsharded():
    import torch.multiprocessing as mp
    import oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B, H, S, d = 2, 3, 48, 32
    rng = np.random.default_rng(11)
    Q, K, V = (rng.standard_normal((B * H, S, d), dtype=np.float32) for _ in range(3))
    full = oracle.attention(Q[None], K[None], V[None], causal=True)[0]
    cat = np.concatenate([g[3] for g in got], axis=0)
    np.testing.assert_array_equal(cat, full)           # sharding by head is bit-exact
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == B * H
    for g in got:
        assert abs(g[4] - float(full.astype(np.float64).sum())) < 1e-6
        assert g[5] == 2.0                             # MAX over ranks of (1 + rank)
