"""Index-range sharding of one MSM over the ranks of a torch.distributed group.

Each rank owns terms [lo, hi) of an n-term MSM, computes one partial sum for its shard and the only
exchange step is an all_gather of the fixed-size partials followed by a local combine on every rank.
The compute is injected (`partial_fn`, `combine_fn`): on GPUs it is zkt_g1_msm_collect's Jacobian
partial + zkt_g1_jac_sum_dev over RCCL (bench.py); the CPU test (tests/test_sharded_gloo.py) injects the
oracle over gloo to check the orchestration itself."""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """contiguous, balanced index range of `rank` (first n % world ranks get one extra term)"""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_sum(partial: torch.Tensor, combine_fn, group=None):
    """all_gather the per-rank partial (any fixed-shape tensor) and combine locally -> combine_fn(stack)"""
    world = dist.get_world_size(group)
    parts = [torch.empty_like(partial) for _ in range(world)]
    dist.all_gather(parts, partial.contiguous(), group=group)
    return combine_fn(torch.stack(parts))
