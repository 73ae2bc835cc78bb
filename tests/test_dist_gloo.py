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


class _OracleEngine:
    """Stand-in for ev2gym_amd.engine.Engine on a box without a GPU: the same calls bench.RolloutLoop makes, served by the
    CPU oracle on the current window of the scenario pool."""

    def __init__(self, pool, E, rk, sk):
        from oracle.oracle import Oracle
        self._Oracle, self.pool, self.E, self.M, self.rk, self.sk = Oracle, pool, E, pool.n_envs, rk, sk
        self.T, self.P = pool.n_steps, pool.n_ports
        self.ora, self.current_step, self.offsets, self.all_acts = None, 0, [], None

    def reset(self, obs=None, offset=0):
        if self.ora is not None:
            self.ora.close()
        self.offsets.append(offset)
        self.ora = self._Oracle(self.pool.select((np.arange(self.E) + offset) % self.M), self.rk, self.sk)
        self.ora.reset()
        self.current_step = 0

    def step_n(self, k, acts, a_stride, obs, o_stride, rew, r_stride, done, d_stride, mask, m_stride, auto_reset=False, persistent=False):
        for i in range(k):   # `acts` is the device address of step current_step's actions, a_stride apart: here the host array
            self.ora.step(self.all_acts[self.current_step].numpy().copy())
            self.current_step += 1

    def last_step_n_kernel_ms(self):
        return 0.0

    def stats(self, out=None):
        import torch
        out.copy_(torch.from_numpy(np.nan_to_num(self.ora.stats())))
        return out


def _loop_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from bench import RolloutLoop
    from ev2gym_amd import _abi
    from ev2gym_amd.dist import AsyncStatsGather
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E, M = 5, 15
    pool = generate(GenConfig.v2g_profit_plus_loads(M, 10, seed=300 + rank))   # every rank draws its own pool, like bench.py
    eng = _OracleEngine(pool, E, 0, 0)
    T, P = eng.T, eng.P
    acts = torch.from_numpy(host_uniform(T * E * P, 50 + rank, -1.0, 1.0).reshape(T, E, P))
    eng.all_acts = acts
    gath = AsyncStatsGather(E, world, "cpu")
    loop = RolloutLoop(eng, E, P, T, M, acts, None, None, None, None, None, gath=gath, actor=None)
    loop.reset()
    loop.run(2 * T + 40, persistent=True)       # two whole episodes and the start of a third, in uneven pieces ...
    loop.run(T - 40, persistent=False)          # ... the third one finished by a second call
    out = gath.finish()
    assert loop.episodes == 3 and gath.collectives == 3 and eng.offsets == [5, 10, 0, 5]
    if rank == 0:
        np.save(os.path.join(out_dir, "loop_gathered.npy"), out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bench_rollout_loop_over_two_ranks(tmp_path):
    """bench.py's stepping loop (episode boundaries inside and between run() calls, statistics -> asynchronous gather ->
    reset onto the next window of the scenario pool, finish()) on two gloo ranks, the CPU oracle standing in for the HIP
    engine: the last gathered block equals what each rank's third episode produces on its own."""
    import torch.multiprocessing as mp
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate
    from oracle.oracle import Oracle
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_loop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "loop_gathered.npy")
    E, M = 5, 15
    want = []
    for rank in range(2):
        pool = generate(GenConfig.v2g_profit_plus_loads(M, 10, seed=300 + rank))
        T, P = pool.n_steps, pool.n_ports
        acts = host_uniform(T * E * P, 50 + rank, -1.0, 1.0).reshape(T, E, P)
        ora = Oracle(pool.select((np.arange(E) + 0) % M), 0, 0)    # third episode: offset (5 + 2*5) % 15 = 0
        ora.reset()
        for t in range(T):
            ora.step(acts[t].copy())
        want.append(np.nan_to_num(ora.stats()))
        ora.close()
    assert np.array_equal(got, np.concatenate(want, 0))


def _full_episode_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time
    import torch
    import torch.distributed as dist
    from bench import RolloutLoop, full_episode_pass
    from ev2gym_amd.dist import AsyncStatsGather
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E, M = 3, 6
    pool = generate(GenConfig.v2g_profit_plus_loads(M, 6, seed=400 + rank))
    eng = _OracleEngine(pool, E, 0, 0)
    if rank == 1:   # a slower rank: on its own clock it would run fewer episodes (and issue fewer gathers) than rank 0
        fast = eng.step_n
        eng.step_n = lambda *a, **k: (time.sleep(0.004), fast(*a, **k))[1]
    T, P = eng.T, eng.P
    eng.all_acts = torch.from_numpy(host_uniform(T * E * P, 60 + rank, -1.0, 1.0).reshape(T, E, P))
    gath = AsyncStatsGather(E, world, "cpu")
    loop = RolloutLoop(eng, E, P, T, M, eng.all_acts, None, None, None, None, None, gath=gath, actor=None)

    def barrier():
        gath.finish()
        dist.barrier()

    def agree_max(n):
        tt = torch.tensor([n], dtype=torch.int64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        with open(os.path.join(out_dir, f"local_{rank}.txt"), "w") as f:
            f.write(str(n))
        return int(tt.item())
    n_ep, ep_s = full_episode_pass(loop, T, barrier, agree_max, min_s=0.05)
    with open(os.path.join(out_dir, f"agreed_{rank}.txt"), "w") as f:
        f.write(f"{n_ep} {loop.episodes} {gath.collectives}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bench_full_episode_pass_runs_the_same_number_of_episodes_on_every_rank(tmp_path):
    """bench.py's whole-episode measurement issues one collective per episode: two ranks of different speed must agree on the
    number of episodes up front (a count taken from each rank's own clock inside the loop leaves one rank waiting forever)."""
    import torch.multiprocessing as mp
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_full_episode_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a0, a1 = [open(tmp_path / f"agreed_{r}.txt").read().split() for r in range(2)]
    assert a0 == a1 and int(a0[0]) >= 3 and int(a0[1]) == int(a0[0]) + 2 == int(a0[2])   # + the warm-up and the timed single episode
    l0, l1 = [int(open(tmp_path / f"local_{r}.txt").read()) for r in range(2)]
    assert int(a0[0]) == max(l0, l1)


def _refusal_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from bench_stub_engine import StubEngine
    from ev2gym_amd import _abi
    from ev2gym_amd.dist import gather_stats_tensor
    from ev2gym_amd.scenario_gen import GenConfig, generate
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pool = generate(GenConfig.v2g_profit_plus_loads(8, 6, seed=500 + rank))
    ids = [StubEngine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    res = {}
    # unequal shards: rank r steps 3 + r envs -- every rank must refuse, none may be left waiting in a collective
    eng = StubEngine(pool, 0, 0, n_active_envs=3 + rank)
    eng.reset(offset=0)
    eng.comm_init(ids[0], rank, world)
    try:
        eng.gather_stats()
        res["unequal"] = "gathered"
    except RuntimeError as ex:
        res["unequal"] = str(ex)
    dist.barrier()   # (reached by both ranks only if neither hangs above)
    eng.close()
    # equal shards: the same rows torch's gather delivers, rank-major
    eng = StubEngine(pool, 0, 0, n_active_envs=4)
    eng.reset(offset=0)
    eng.comm_init(ids[0], rank, world)
    got = eng.gather_stats()
    want = gather_stats_tensor(torch.from_numpy(np.nan_to_num(eng.ora.stats())))
    res["equal_ok"] = bool(torch.equal(got, want)) and tuple(got.shape) == (world * 4, _abi.N_STATS) and eng.comm_world_size == world
    eng.close()
    import json
    with open(os.path.join(out_dir, f"refusal_{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_c_abi_gather_protocol_refuses_unequal_shards_on_every_rank(tmp_path):
    """ev2g_gather_stats (csrc/ev2g_host.hip) all-gathers the ranks' env counts before the statistics and
    refuses unequal shards instead of writing rows to wrong offsets.  The stand-in engine restates that protocol over gloo (RCCL with more
    than one rank has never run on hardware available to the build): with shards of 3 and 4 envs BOTH ranks get the error that names the
    offending rank -- and both reach the barrier behind it --, with equal shards the rows equal torch's own gather."""
    import json
    import torch.multiprocessing as mp
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_refusal_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [json.load(open(tmp_path / f"refusal_{r}.json")) for r in range(2)]
    assert "needs equal shards" in r0["unequal"] and "rank 1 steps 4 envs, this rank 3" in r0["unequal"], r0
    assert "needs equal shards" in r1["unequal"] and "rank 0 steps 3 envs, this rank 4" in r1["unequal"], r1
    assert r0["equal_ok"] and r1["equal_ok"]
