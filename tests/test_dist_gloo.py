"""N > 1 path on CPU: two processes (gloo), envs sharded contiguously, episode statistics all-gathered.
The stepping itself is done by the CPU oracle here (no GPU in this tier); what is under test is the sharding
arithmetic and the collective, which are backend-agnostic (RCCL on the GPU box)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from ev2gym_amd import _abi
    from ev2gym_amd.dist import env_range, gather_stats_tensor
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E = 13   # deliberately not divisible by the world size
    full = generate(GenConfig.v2g_profit_plus_loads(E, 10, seed=21))
    shard = full.shard(rank, world)
    lo, hi = env_range(E, rank, world)
    assert shard.n_envs == hi - lo
    ora = Oracle(shard, 0, 0)
    ora.reset()
    T, P = full.n_steps, full.n_ports
    for t in range(T):
        a = host_uniform(E * P, 100 + t, -1.0, 1.0).reshape(E, P)[lo:hi].copy()   # the global action stream, sliced
        ora.step(a)
    st = torch.from_numpy(ora.stats())
    allst = gather_stats_tensor(st)
    assert allst.shape == (E, _abi.N_STATS)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), allst.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_plus_gather_equal_single_process(tmp_path):
    import torch.multiprocessing as mp
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate
    from oracle.oracle import Oracle
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    E = 13
    full = generate(GenConfig.v2g_profit_plus_loads(E, 10, seed=21))
    ora = Oracle(full, 0, 0)
    ora.reset()
    for t in range(full.n_steps):
        ora.step(host_uniform(E * full.n_ports, 100 + t, -1.0, 1.0).reshape(E, full.n_ports))
    want = ora.stats()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(want)), "sharded + gathered statistics must equal the single-process run bit for bit"


def _async_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from ev2gym_amd import _abi
    from ev2gym_amd.dist import AsyncStatsGather
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E = 6
    g = AsyncStatsGather(E, world, "cpu")
    seen = []
    for ep in range(5):   # five "episodes": the buffers alternate, every gather must carry that episode's values
        buf = g.buffer()
        buf.copy_(torch.full((E, _abi.N_STATS), float(100 * ep + rank), dtype=torch.float64)
                  + torch.arange(E, dtype=torch.float64)[:, None])
        g.launch()
        if ep >= 1:       # the previous episode's result is complete once its work was waited on in buffer()/finish()
            pass
    out = g.finish()
    seen.append(out.clone())
    want = torch.cat([torch.full((E, _abi.N_STATS), float(100 * 4 + r), dtype=torch.float64)
                      + torch.arange(E, dtype=torch.float64)[:, None] for r in range(world)], 0)
    assert torch.equal(out, want), (rank, out[:, 0], want[:, 0])
    # the other receive buffer still holds the episode before (nothing overwrote it out of order)
    prev = g.recv[g.last ^ 1]
    wantp = torch.cat([torch.full((E, _abi.N_STATS), float(100 * 3 + r), dtype=torch.float64)
                       + torch.arange(E, dtype=torch.float64)[:, None] for r in range(world)], 0)
    assert torch.equal(prev, wantp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_async_double_buffered_stats_gather_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_async_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
