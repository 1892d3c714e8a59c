"""Multi-GPU sharding of the path: one process per GPU, torch.distributed (backend "nccl" is RCCL
on ROCm; "gloo" for the CPU rehearsal).

What shards and what is exchanged
  * NTT batches, keygen and sign are independent per polynomial / signature: ranks own contiguous
    blocks, no data-path collective.
  * aggregate() is a sum over signatures and verify()'s target a sum over signers: each rank sums
    its block exactly into int64 (fz_aggregate_partial / fz_target_partial), then ONE all-reduce
    (ncclSum on int64; sums of centred int32 products need > 32 bits) and a centring kernel
    (fz_reduce_i64).  Integer sums are associative, so the result is bit-identical for any number
    of ranks and any reduction order.

The functions take the per-rank partial computation as a callable so that the same code is
exercised on CPU (gloo, partials from the oracle -- tests/test_dist_cpu.py) and on GPUs (partials
from the HIP kernels -- bench.py).
"""
from typing import Callable, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` items owned by `rank`: the first total % world ranks
    get one extra item.  Empty blocks are allowed (total < world)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_sum_i64(t, group=None):
    """In-place sum of an int64 tensor over the ranks of `group`; a no-op without an initialised
    process group (single-GPU runs)."""
    import torch
    import torch.distributed as dist
    if t.dtype != torch.int64:
        raise TypeError("partial sums must be int64: 8 centred int32 values already overflow int32")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def sharded_sum(partial_fn: Callable[[int, int], "object"], total: int, rank: int, world: int, group=None):
    """partial_fn(lo, hi) -> int64 tensor holding this rank's exact partial sums over items
    [lo, hi) (zeros for an empty block); returns the all-reduced tensor."""
    lo, hi = shard_range(total, rank, world)
    return allreduce_sum_i64(partial_fn(lo, hi), group)
