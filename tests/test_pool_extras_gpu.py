"""Round-2 surface of the C-ABI (include/ev2g.h v2), through the C-ABI on the GPU:
  * the device-resident scenario pool: E envs running a window of M >= E scenarios, chosen per reset (ev2g_reset_ex) --
    the per-episode scenario draw of EV2Gym.reset() (ev2gym_env.py:243-296) without a host round trip;
  * the fused cost_function output (rl_agent/cost.py:8-27) against the reference fixtures;
  * float32 actions in / float32 observations out (the policy-network interface);
  * the RCCL statistics gather executed on the device (world size 1).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-9
CFG = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "ev2gym_amd", "example_config_files")


def _close(a, b, what, tol=RTOL):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert (np.isnan(a) == np.isnan(b)).all(), f"{what}: NaN pattern differs"
    err = np.nan_to_num(np.abs(a - b) / np.maximum(1.0, np.abs(np.nan_to_num(b))))
    assert err.max(initial=0.0) <= tol, f"{what}: max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


def _shape(kind, M, seed):
    from ev2gym_amd import _abi
    from ev2gym_amd.scenario_gen import GenConfig, generate
    RK, SK = _abi.REWARD_KINDS, _abi.STATE_KINDS
    if kind == "wave_v2gppl":     # ev2g_step_wave
        return generate(GenConfig.v2g_profit_plus_loads(M, 50, 1, seed=seed)), RK["ProfitMax_TrPenalty_UserIncentives"], SK["V2G_profit_max_loads"], -1.0
    if kind == "wave_pst":        # ev2g_step_wave, three envs per wavefront
        return generate(GenConfig.public_pst(M, 20, seed=seed)), RK["SquaredTrackingErrorReward"], SK["PublicPST"], 0.0
    if kind == "v2_multi":        # ev2g_step_v2<256>: multi-port chargers, several transformers
        return (generate(GenConfig.v2g_profit_plus_loads(M, 40, 3, seed=seed, number_of_ports_per_cs=2)),
                RK["ProfitMax_TrPenalty_UserIncentives"], SK["V2G_profit_max_loads"], -1.2)
    if kind == "generic":         # ev2g_step_kernel (P > 1024), multi-port chargers whose ports straddle wavefronts
        return (generate(GenConfig.v2g_profit_plus_loads(M, 400, 2, seed=seed, number_of_ports_per_cs=3)),
                RK["ProfitMax_TrPenalty_UserIncentives"], SK["V2G_profit_max_loads"], -1.3)
    if kind == "topo_het":        # ev2g_step_kernel: a topology file's chargers, each with its own port count / limits / voltage / phases
        n_ports = np.array([4, 3, 3, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1])
        C = len(n_ports)
        topo = dict(n_ports=n_ports, transformer=np.arange(C) % 3, min_charge_current=np.full(C, 6.0),
                    max_charge_current=np.where(np.arange(C) % 2, 16.0, 32.0), min_discharge_current=np.zeros(C),
                    max_discharge_current=np.where(np.arange(C) % 2, -16.0, -32.0), voltage=np.where(np.arange(C) % 3, 400.0, 230.0),
                    phases=np.where(np.arange(C) % 4 == 1, 1, 3), tr_max_power=np.array([90.0, 60.0, 45.0]))
        return (generate(GenConfig.v2g_profit_plus_loads(M, seed=seed, topology=topo, spawn_multiplier=9.0)),
                RK["ProfitMax_TrPenalty_UserIncentives"], SK["V2G_profit_max_loads"], -1.3)
    raise ValueError(kind)


@pytest.mark.parametrize("kind,M,E,K", [("wave_v2gppl", 41, 12, 112), ("wave_pst", 50, 17, 112), ("v2_multi", 23, 9, 112),
                                        ("generic", 7, 3, 30), ("topo_het", 37, 21, 112)])
def test_scenario_pool_window_matches_oracle(kind, M, E, K):
    """E envs stepping a window of an M-scenario pool == the oracle on exactly those scenarios, for windows that start
    anywhere in the pool (including ones that wrap around its end); statistics and peeks follow the window."""
    from ev2gym_amd.engine import Engine, host_uniform
    from oracle.oracle import Oracle
    pool, rk, sk, lo = _shape(kind, M, seed=5)
    eng = Engine(pool, rk, sk, device=0, flags=4, n_active_envs=E)
    assert (eng.E, eng.M) == (E, M)
    if kind == "topo_het":
        assert eng.kernel_name == "ev2g_step_kernel" and "port counts" in eng.fallback_reason and eng.P == 25
    P, D, T = eng.P, eng.D, eng.T
    K = min(K, T)
    d_act = eng.empty((K, E, P))
    eng.fill_uniform(d_act, K * E * P, 21, lo, 1.0)
    acts = host_uniform(K * E * P, 21, lo, 1.0).reshape(K, E, P)
    d_obs, d_rew, d_mask = eng.empty((K, E, D)), eng.empty((K, E)), eng.empty((K, E, P), np.uint8)
    d_obs0 = eng.empty((E, D))
    for off in (0, M - E // 2, 3 * M + 5):       # a plain window, one that wraps, an offset beyond M (taken modulo)
        ids = (np.arange(E) + off) % M
        ora = Oracle(pool.select(ids), rk, sk)
        eng.reset(d_obs0, offset=off)
        assert eng.scenario_offset == off % M
        _close(d_obs0.to_host(), ora.reset(), f"reset obs, offset {off}")
        eng.step_n(K, d_act, E * P, d_obs, E * D, d_rew, E, None, 0, d_mask, E * P, auto_reset=False, persistent=(off != 0))
        obs, rew, mask = d_obs.to_host(), d_rew.to_host(), d_mask.to_host()
        faulted = False
        for t in range(K):
            o, r, d, m, rc = ora.step(acts[t].copy())
            faulted = faulted or rc != 0
            assert np.array_equal(mask[t], m), f"mask[{t}], offset {off}"
            _close(obs[t], o, f"obs[{t}], offset {off}")
            _close(rew[t], r, f"reward[{t}], offset {off}")
        if K == T:
            _close(eng.stats(), ora.stats(), f"episode statistics, offset {off}")
        for e in (0, E - 1):
            pk, po = eng.peek(e), ora.peek(e)
            _close(pk["port_capacity"], po["cap"], "capacity")
            assert (pk["port_session"] == po["session"]).all()
        # the reference raises when the running sum of a charger's currents passes its limit (ev_charger.py:203-205; with several V2G ports
        # and actions beyond the box that happens): the engine flags exactly the runs the oracle flags
        if faulted:
            with pytest.raises(Exception, match="over-current"):
                eng.check_faults()
        else:
            eng.check_faults()
        ora.close()
    eng.close()


@pytest.mark.parametrize("kind,M,E", [("wave_v2gppl", 30, 8), ("v2_multi", 20, 6), ("topo_het", 19, 7)])
@pytest.mark.parametrize("persistent", [True, False])
def test_auto_reset_moves_to_the_next_scenarios_of_the_pool(kind, M, E, persistent):
    """ev2g_step_n(auto_reset = NEXT): when the episode ends inside a fused run -- inside ONE persistent launch too -- the
    envs continue on the next E scenarios of the pool, not on the ones they just finished."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, host_uniform
    from oracle.oracle import Oracle
    pool, rk, sk, lo = _shape(kind, M, seed=9)
    eng = Engine(pool, rk, sk, device=0, flags=4, n_active_envs=E)
    P, D, T = eng.P, eng.D, eng.T
    K = 2 * T + 10
    off0 = M - 3
    d_act = eng.empty((K, E, P))
    eng.fill_uniform(d_act, K * E * P, 4, lo, 1.0)
    acts = host_uniform(K * E * P, 4, lo, 1.0).reshape(K, E, P)
    d_obs, d_rew, d_done = eng.empty((K, E, D)), eng.empty((K, E)), eng.empty((K, E), np.uint8)
    eng.reset(offset=off0)
    eng.step_n(K, d_act, E * P, d_obs, E * D, d_rew, E, d_done, E, None, 0, auto_reset=_abi.AUTO_RESET_NEXT, persistent=persistent)
    assert eng.scenario_offset == (off0 + 2 * E) % M and eng.current_step == 10
    obs, rew, done = d_obs.to_host(), d_rew.to_host(), d_done.to_host()
    k = 0
    for ep in range(3):
        ids = (np.arange(E) + off0 + ep * E) % M
        ora = Oracle(pool.select(ids), rk, sk)
        ora.reset()
        for t in range(T if ep < 2 else 10):
            o, r, d, m, rc = ora.step(acts[k].copy())
            _close(obs[k], o, f"episode {ep} obs[{t}]")
            _close(rew[k], r, f"episode {ep} reward[{t}]")
            assert (done[k] == d).all()
            k += 1
        ora.close()
    eng.close()


@pytest.mark.parametrize("name", ["v2gppl_rand_s2", "v2gppl_c60r5_rand_s13", "v2gppl_p2_rand_s11", "pst_rand_s2"])
@pytest.mark.parametrize("cost_kind", [1, 2])
def test_fused_cost_matches_reference_fixture(name, cost_kind):
    """cost_function in the batched API (ev2gym_env.py:434-438, rl_agent/cost.py:8-27): the value the reference's cost
    plugins return at every step of a reference episode, recomputed from the fixture's own transformer overloads,
    departure scores and charger profits."""
    from ev2gym_amd.engine import Engine
    z, batch, rk, sk = load_golden(os.path.join(GOLDEN_DIR, name + ".npz"))
    eng = Engine(batch, rk, sk, device=0, flags=4, cost_kind=cost_kind)
    E, P, D = eng.E, eng.P, eng.D
    d_act, d_obs, d_rew, d_cost = eng.empty((E, P)), eng.empty((E, D)), eng.empty((E,)), eng.empty((E,))
    eng.set_extras(cost=d_cost)
    eng.reset(d_obs)
    prev = np.zeros(z["trj_cs_profits"].shape[1])
    for t in range(len(z["act"])):
        d_act.upload(z["act"][t:t + 1])
        eng.step(d_act, d_obs, d_rew, None, None)
        if cost_kind == 1:
            want = 0.0
            for ov in z["trj_tr_overload"][t]:
                want += 100 * ov
            for sc in z["trj_dep_score"][t][:int(z["trj_n_departed"][t])]:
                want += 100 * np.exp(-10 * sc)
        else:
            want = float((z["trj_cs_profits"][t] - prev).sum())
            prev = z["trj_cs_profits"][t]
        _close(d_cost.to_host()[0], want, f"cost[{t}]")
        _close(d_rew.to_host()[0], z["trj_reward"][t], f"reward[{t}]")
    eng.close()


@pytest.mark.parametrize("kind", ["wave_v2gppl", "wave_pst", "v2_multi", "wave_v2gppl:pst_V2G_profitmaxV2", "wave_pst:SquaredTrackingErrorRewardWithPenalty"])
def test_float32_actions_and_observations(kind):
    """The policy-network interface: float32 actions are widened on entry exactly like float64 ones holding the same
    values, and the float32 observation is the float64 one rounded once."""
    from ev2gym_amd.engine import Engine, host_uniform
    from ev2gym_amd import _abi
    E = 33
    kind, _, reward = kind.partition(":")
    pool, rk, sk, lo = _shape(kind, E, seed=3)
    if reward:   # the run-time-selected reward instantiation of the fast path, float32 flavour
        rk = _abi.REWARD_KINDS[reward]
    P, T = pool.n_ports, pool.n_steps
    K = 60
    a32 = host_uniform(K * E * P, 8, lo, 1.0).astype(np.float32).reshape(K, E, P)
    outs = []
    for f32 in (False, True):
        eng = Engine(pool, rk, sk, device=0, flags=4)
        D = eng.D
        d_obs, d_rew = eng.empty((K, E, D)), eng.empty((K, E))
        d_obs32 = eng.empty((K, E, D), np.float32)
        if f32:
            d_act = eng.empty((K, E, P), np.float32).upload(a32)
            eng.set_extras(obs_f32=d_obs32, obs_f32_stride=E * D, actions_f32=d_act)
            # the extras' step strides count from the start of each ev2g_step_n run: one run per launch mode, continuing one episode
            eng.reset()
            eng.step_n(25, None, E * P, d_obs, E * D, d_rew, E, auto_reset=False, persistent=True)
            first = (d_obs.to_host()[:25].copy(), d_rew.to_host()[:25].copy(), d_obs32.to_host()[:25].copy())
            eng.set_extras(obs_f32=d_obs32.at(25 * E * D), obs_f32_stride=E * D, actions_f32=d_act.at(25 * E * P))
            eng.step_n(K - 25, None, E * P, d_obs.at(25 * E * D), E * D, d_rew.at(25 * E), E, auto_reset=False, persistent=False)
            o, r, o32 = d_obs.to_host(), d_rew.to_host(), d_obs32.to_host()
            assert np.array_equal(o[:25], first[0]) and np.array_equal(o32[:25], first[2])
            outs.append((o, r, o32))
        else:
            d_act = eng.empty((K, E, P)).upload(a32.astype(np.float64))
            eng.step_n(K, d_act, E * P, d_obs, E * D, d_rew, E, auto_reset=False, persistent=True)
            outs.append((d_obs.to_host(), d_rew.to_host(), None))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[1][2], outs[0][0].astype(np.float32)), "obs_f32 must be the float64 observation rounded once"


def test_charger_histories_on_the_fast_path():
    """EV2G_FLAG_LOG_CS_HISTORY no longer leaves the fast path: cs_power / cs_current of every step (ev2gym_env.py:533-535)
    from ev2g_step_wave == from the general kernel, which the reference fixtures pin."""
    from ev2gym_amd.engine import Engine, host_uniform
    E = 19
    pool, rk, sk, lo = _shape("wave_v2gppl", E, seed=12)
    res = []
    for forced in (None, "v2"):
        if forced:
            os.environ["EV2G_KERNEL"] = forced
        try:
            eng = Engine(pool, rk, sk, device=0, flags=1 | 4)
        finally:
            os.environ.pop("EV2G_KERNEL", None)
        assert eng.kernel_name.startswith("ev2g_step_v2" if forced else "ev2g_step_wave")
        P, T = eng.P, eng.T
        d_act = eng.empty((T, E, P))
        eng.fill_uniform(d_act, T * E * P, 2, lo, 1.0)
        eng.step_n(40, d_act, E * P, auto_reset=False, persistent=True)
        snap = []
        for k in range(40, T):
            eng.step(d_act.at(k * E * P))
            if k % 9 == 0:
                snap.append([eng.peek(e) for e in (0, E - 1)])
        res.append(snap)
        eng.close()
    for sa, sb in zip(*res):
        for pa, pb in zip(sa, sb):
            for key in ("cs_power", "cs_amps", "cs_profits", "cs_energy_charged", "cs_energy_discharged"):
                _close(pa[key], pb[key], key, tol=1e-12)


def test_vec_env_draws_fresh_scenarios_every_episode():
    """EV2GymVec.reset() without a seed moves to other scenarios of the resident pool (ev2gym_env.py:243-296 draws a new
    scenario per reset); reset(seed=s) is reproducible; the auto-resetting SB3 adapter uses the same draw."""
    from ev2gym_amd.sb3_vec_env import EV2GymSB3VecEnv
    from ev2gym_amd.vec_env import EV2GymVec
    env = EV2GymVec(config_file=os.path.join(CFG, "V2GProfitPlusLoads.yaml"), num_envs=32, seed=3, pool_factor=8,
                    state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives", use_torch=False)
    assert env.engine.M == 8 * 32 and env.engine.E == 32
    a = env.full_like_actions(0.7)

    def episode():
        masks, rews = [], []
        for _ in range(env.simulation_length):
            obs, rew, done, trunc, info = env.step(a)
            masks.append(np.asarray(info["action_mask"]).copy())
            rews.append(np.asarray(rew).copy())
        return np.stack(masks), np.stack(rews)
    env.reset(seed=11)
    off11 = env.engine.scenario_offset
    m1, r1 = episode()
    env.reset()
    m2, r2 = episode()
    env.reset()
    m3, r3 = episode()
    assert not np.array_equal(m1, m2) and not np.array_equal(m2, m3) and not np.array_equal(r1, r3)
    env.reset(seed=11)
    assert env.engine.scenario_offset == off11
    m4, r4 = episode()
    assert np.array_equal(m1, m4) and np.array_equal(r1, r4), "reset(seed=s) must reproduce the episode"
    # every env of a batch runs a different scenario
    assert len({m1[:, e].tobytes() for e in range(32)}) == 32
    env.close()
    venv = EV2GymSB3VecEnv(config_file=os.path.join(CFG, "V2GProfitPlusLoads.yaml"), num_envs=16, seed=1,
                           state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives")
    venv.reset()
    T = venv.vec.simulation_length
    acts = np.full((16, venv.vec.number_of_ports), 0.5, np.float32)
    seen = []
    for ep in range(3):
        ms = []
        for t in range(T):
            obs, rew, done, infos = venv.step(acts)
            ms.append(np.stack([i["action_mask"] for i in infos]))
        assert done.all() and "terminal_observation" in infos[0]
        seen.append(np.stack(ms))
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])
    venv.close()


def test_rccl_statistics_gather_runs_on_the_device():
    """The multi-GPU exchange of the path (one all-gather of the [E,17] statistics per episode) executed
    through RCCL on this GPU: a world-size-1 `nccl` process group, AsyncStatsGather on a non-default stream and
    EV2GymVec.get_statistics_all_ranks()."""
    import torch
    import torch.distributed as dist
    from ev2gym_amd import _abi
    from ev2gym_amd.dist import AsyncStatsGather, gather_stats_tensor
    from ev2gym_amd.vec_env import EV2GymVec
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        stream = torch.cuda.Stream(device=0)
        with torch.cuda.stream(stream):
            env = EV2GymVec(config_file=os.path.join(CFG, "V2GProfitPlusLoads.yaml"), num_envs=48, seed=2, pool_factor=2,
                            state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives",
                            use_torch=True, rank=0, world_size=1)
            a = env.full_like_actions(1.0)
            for _ in range(env.simulation_length):
                env.step(a)
            local = env.engine.stats()
            allr = env.get_statistics_all_ranks()
            assert tuple(allr.shape) == (48, _abi.N_STATS)
            assert np.array_equal(np.nan_to_num(allr.cpu().numpy()), np.nan_to_num(local))
            # the asynchronous, double-buffered gather bench.py and the training loops use: with a process group up it
            # issues ncclAllGather on RCCL even at world size 1
            g = AsyncStatsGather(48, 1, torch.device("cuda", 0))
            for ep in range(3):
                env.engine.stats(out=g.buffer())
                g.launch()
            got = g.finish()
            assert g.collectives == 3
            assert np.array_equal(np.nan_to_num(got.cpu().numpy()), np.nan_to_num(local))
            tot = torch.nan_to_num(got).sum(dim=0)
            dist.all_reduce(tot)
            torch.cuda.synchronize()
            assert torch.equal(tot, torch.nan_to_num(got).sum(dim=0))
            assert torch.equal(torch.nan_to_num(gather_stats_tensor(got)), torch.nan_to_num(got))
            env.close()
    finally:
        dist.destroy_process_group()


def test_c_abi_rccl_gather_for_hosts_without_torch():
    """ev2g_comm_get_unique_id / ev2g_comm_init / ev2g_gather_stats: the statistics all-gather issued by the library itself
    through librccl on the engine's stream (world size 1 on this box; bench.py --gpus N runs it across ranks)."""
    from ev2gym_amd.engine import Engine, EngineError
    pool, rk, sk, lo = _shape("wave_v2gppl", 24, seed=3)
    eng = Engine(pool, rk, sk, device=0, flags=4, n_active_envs=16)
    E, P, T = eng.E, eng.P, eng.T
    d_act = eng.empty((T, E, P))
    eng.fill_uniform(d_act, T * E * P, 8, lo, 1.0)
    eng.reset(offset=5)
    eng.step_n(T, d_act, E * P, None, 0, None, 0, None, 0, None, 0, auto_reset=False, persistent=True)
    with pytest.raises(EngineError, match="communicator"):
        eng.gather_stats()
    uid = Engine.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(uid, 0, 1)
    assert eng.comm_world_size == 1
    want = eng.stats()
    got = eng.gather_stats()
    assert got.shape == (E, 17) and np.array_equal(got, want, equal_nan=True) and eng.comm_gathers == 1
    out = eng.empty((E, 17))
    eng.gather_stats(out)                       # device output, stream-ordered
    assert np.array_equal(out.to_host(), want, equal_nan=True) and eng.comm_gathers == 2
    eng.close()
