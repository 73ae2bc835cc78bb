"""Multi-GPU: one process per GPU, envs sharded contiguously, no data-path collective.

The reference is a single-process single-env simulator (no collectives anywhere).
Envs are independent, so the only exchange is the per-episode statistics block [E_local, 17] float64, gathered
with one RCCL all-gather over xGMI (`torch.distributed`, backend "nccl" == RCCL on ROCm; "gloo" on CPU for
tests).  Per-step traffic is zero.
"""
from __future__ import annotations

import numpy as np

from . import _abi


def env_range(n_envs: int, rank: int, world: int):
    """Contiguous [lo, hi) env range of `rank` (same split as ScenarioBatch.shard)."""
    return rank * n_envs // world, (rank + 1) * n_envs // world


def gather_stats_tensor(stats, group=None):
    """all_gather of a [E_local, 17] tensor -> [sum E_local, 17] (ranks may own different env counts)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return stats
    world = dist.get_world_size(group)
    n = torch.tensor([stats.shape[0]], dtype=torch.int64, device=stats.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0], stats.shape[1]), dtype=stats.dtype, device=stats.device)
        dist.all_gather_into_tensor(out, stats.contiguous(), group=group)
        return out
    m = max(counts)
    pad = torch.zeros((m, stats.shape[1]), dtype=stats.dtype, device=stats.device)
    pad[:stats.shape[0]] = stats
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


def gather_stats(vec_env, group=None):
    """Episode statistics of all ranks for an EV2GymVec (device tensor in, device tensor out when torch is used)."""
    import torch
    if vec_env._torch is not None:
        st = torch.empty((vec_env.num_envs, _abi.N_STATS), dtype=torch.float64, device=f"cuda:{vec_env.device}")
        vec_env.engine.stats(out=st)
    else:
        st = torch.from_numpy(np.ascontiguousarray(vec_env.engine.stats()))
    return gather_stats_tensor(st, group)
