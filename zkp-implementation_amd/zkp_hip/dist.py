"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm); gloo on CPU for tests.

MSM shards by contiguous point/scalar chunk.  The only exchange is the final one: every rank's
partial sum (one extended-Jacobian point, 24 x u64 = 192 B) is all-gathered and the ranks add the partials locally
-- EC addition is not an RCCL reduction operator, so the "all-reduce of partial sums" is gather + local add.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous chunk [lo, hi) of n terms owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_partials(partial, group=None, device=None):
    """partial: (24,) uint64 numpy (X, Y, ZZ, ZZZ).  Returns (world, 24) uint64 numpy on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.ascontiguousarray(partial, dtype=np.uint64).reshape(1, 24)
    world = dist.get_world_size(group)
    mine = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    out = torch.empty(world * 24, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().view(np.uint64).reshape(world, 24)


def msm_g1_sharded(zkp, bases_local, scalars_local, n_local, group=None, device=None, stream=None):
    """Every rank passes ITS chunk (bases resident on its GPU); every rank returns the full MSM (affine, is_inf)."""
    part = zkp.msm_g1_partial_dev(bases_local, scalars_local, n_local, stream=stream)
    parts = allgather_partials(part, group=group, device=device)
    return zkp.g1_xyzz_sum(parts)
