"""Batch x head sharding across the GPUs of one node.

Each (b, h) pair is an independent attention problem, so the forward pass shards with NO data-path
collective: rank r of N owns the contiguous range of flattened heads g = b*H + h in
[r*BH/N, (r+1)*BH/N), i.e. one contiguous slab of every dense [B,H,S,d] tensor, and calls the
single-GPU flash_attention on it.  torch.distributed (RCCL over xGMI on the GPU box, gloo in the
CPU tests) is used only outside the timed region: MAX of per-rank elapsed time, SUM of checksums.
"""
from __future__ import annotations


def shard_heads(total_heads: int, rank: int, world: int) -> tuple[int, int]:
    """[lo, hi) of flattened heads owned by `rank`; ranges tile [0, total_heads) exactly."""
    if world <= 0 or not (0 <= rank < world) or total_heads < 0:
        raise ValueError("bad shard arguments")
    return (total_heads * rank) // world, (total_heads * (rank + 1)) // world


def slab(total_heads: int, rank: int, world: int, seq_len: int, d_head: int) -> tuple[int, int]:
    """(element offset, element count) of the rank's slab inside a dense [B*H, S, d] tensor."""
    lo, hi = shard_heads(total_heads, rank, world)
    return lo * seq_len * d_head, (hi - lo) * seq_len * d_head


def reduce_max(value: float, group=None) -> float:
    """MAX over ranks of a host scalar (per-rank elapsed time -> job time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def reduce_sum(value: float, group=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
