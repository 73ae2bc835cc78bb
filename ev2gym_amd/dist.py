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


class AsyncStatsGather:
    """Per-episode statistics all-gather that overlaps with the next episode (equal env counts per rank).

    Two [E,17] send buffers and two [world*E,17] receive buffers alternate.  `buffer()` hands out the send buffer of
    the current episode (first waiting, on the device, for the gather that used it two episodes ago); `launch()`
    starts the all-gather asynchronously -- the collective is ordered after the kernels already queued on the
    current stream, and the current stream does not wait for it, so the next episode's step kernels run while the
    statistics travel over xGMI; `finish()` waits for everything outstanding and returns the last result.
    """

    def __init__(self, n_envs: int, world: int, device, dtype=None, group=None, mode=None):
        import os
        import torch
        dtype = dtype or torch.float64
        self.group, self.world = group, world
        # WHEN the collective is handed to the GPU (EV2G_GATHER_MODE / `mode`):
        #   "deferred" (default): launch() only marks the buffer; the caller calls flush() right AFTER it has enqueued the next episode's step
        #              kernel.  A persistent step kernel occupies every wavefront slot of the chip (1024 workgroups on 256 CUs x 4): a collective
        #              kernel that reaches the GPU first takes a slot and the displaced workgroup starts only when it leaves -- the whole
        #              episode ends that much later (measured: 366 -> 466 us per launch).  Submitted behind the step kernel, the collective
        #              waits for a slot instead and runs next to the following statistics kernel.
        #   "overlap": issued inside launch(), asynchronously (rounds 2-4);  "inline": issued inside launch(), the current stream waits for it.
        self.mode = mode or os.environ.get("EV2G_GATHER_MODE", "deferred")
        self._pending = None
        self.send = [torch.zeros((n_envs, _abi.N_STATS), dtype=dtype, device=device) for _ in range(2)]
        self.recv = [torch.zeros((world * n_envs, _abi.N_STATS), dtype=dtype, device=device) for _ in range(2)]
        self.work = [None, None]
        self.i = 0
        self.launched = 0
        self.collectives = 0   # all-gathers actually issued on the process group

    def buffer(self):
        if self._pending == self.i:
            self.flush()
        w = self.work[self.i]
        if w is not None:
            w.wait()
            self.work[self.i] = None
        return self.send[self.i]

    def _issue(self, i):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():   # also at world size 1: the same RCCL call, one rank
            w = dist.all_gather_into_tensor(self.recv[i], self.send[i], group=self.group, async_op=(self.mode != "inline"))
            self.work[i] = w if self.mode != "inline" else None
            self.collectives += 1
        else:
            self.recv[i].copy_(self.send[i])

    def launch(self):
        self.flush()   # (at most one deferred collective: an episode end without a step kernel behind it)
        if self.mode == "deferred":
            self._pending = self.i
        else:
            self._issue(self.i)
        self.last = self.i
        self.i ^= 1
        self.launched += 1

    def flush(self):
        """"deferred" mode: hand the marked collective to the GPU now (the caller has just enqueued the next step kernel); a no-op otherwise."""
        if self._pending is not None:
            i, self._pending = self._pending, None
            self._issue(i)

    def finish(self):
        self.flush()
        for k in (0, 1):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
        return self.recv[self.last] if self.launched else None
