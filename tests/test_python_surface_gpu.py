"""The reference's Python surface on top of the HIP engine: single-env facade (object graph, fused and host-side
plugins), vectorised env, heuristics.  Checked against the reference goldens and the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "ev2gym_amd", "example_config_files")


def _window(env):
    """The scenarios the envs of an EV2GymVec are running right now: a window of its resident pool."""
    e = env.engine
    return env.scenarios.select((np.arange(e.E) + e.scenario_offset) % e.M)


def _close(a, b, what, tol=1e-9):
    a, b = np.asarray(a, float), np.asarray(b, float)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert err.max(initial=0.0) <= tol, f"{what}: {err.max():.3e}"


@pytest.mark.parametrize("name", ["v2gppl_rand_s2", "pst_rand_s2", "v2gmax_het_rand_s3", "v2gppl_p2_rand_s11",
                                  "v2gppl_c10r3_mixed_s14", "topo_v2gppl_het_rand_s41", "topo_pst_het_mixed_s43"])
@pytest.mark.parametrize("host_plugins", [False, True], ids=["fused", "host_plugins"])
def test_facade_reproduces_reference_episode(name, host_plugins):
    """EV2Gym facade == reference trajectory, with the plugins fused in the kernel and evaluated on the host."""
    from ev2gym_amd.env import EV2Gym
    from ev2gym_amd.rl_agent import reward as R, state as S
    z, batch, rk, sk = load_golden(os.path.join(GOLDEN_DIR, name + ".npz"))
    sf = getattr(S, str(z["case"][2]))
    rf = getattr(R, str(z["case"][3]))
    if host_plugins:   # strip the fusion marker: these become "user-defined" callables evaluated through the facade
        sf0, rf0 = sf, rf
        sf = lambda env, *a: sf0(env, *a)  # noqa: E731
        rf = lambda env, *a: rf0(env, *a)  # noqa: E731
    env = EV2Gym(scenario=batch, state_function=sf, reward_function=rf)
    obs, _ = env.reset()
    _close(obs, z["trj_obs"][0], "reset obs")
    for t in range(len(z["act"])):
        a = z["act"][t].copy()
        obs, rew, done, trunc, info = env.step(a)
        assert (a == z["trj_act_after"][t]).all(), "empty-port actions are zeroed in the caller's array"
        _close(obs, z["trj_obs"][t + 1], f"obs[{t}]")
        _close(rew, z["trj_reward"][t], f"reward[{t}]")
        assert done == bool(z["trj_done"][t]) and trunc is False
        assert (info["action_mask"] == z["trj_mask"][t]).all()
        assert len(env.departing_evs) == z["trj_n_departed"][t]
    assert done
    for k in ("total_ev_served", "total_profits", "total_energy_charged", "total_transformer_overload", "total_reward"):
        g = z["trj_stats"][__import__("ev2gym_amd")._abi.STAT_NAMES.index(k)]
        _close(info[k], g, k)
    with pytest.raises(AssertionError):
        env.step(z["act"][0].copy())
    env.close()


def test_facade_object_graph_and_heuristics():
    from ev2gym_amd.baselines.heuristics import ChargeAsFastAsPossible
    from ev2gym_amd.env import EV2Gym
    z, batch, rk, sk = load_golden(os.path.join(GOLDEN_DIR, "v2gppl_ones_s1.npz"))
    env = EV2Gym(scenario=batch, state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives")
    agent = ChargeAsFastAsPossible()
    for t in range(40):
        obs, rew, done, _, info = env.step(agent.get_action(env))
        _close(rew, z["trj_reward"][t], f"reward[{t}]")
        for i, cs in enumerate(env.charging_stations):
            for j, ev in enumerate(cs.evs_connected):
                p = i * cs.n_ports + j
                if ev is None:
                    assert np.isnan(z["trj_cap"][t, p])
                else:
                    _close(ev.current_capacity, z["trj_cap"][t, p], "ev.current_capacity")
                    _close(ev.get_soc() * ev.battery_capacity, z["trj_cap"][t, p], "soc")
                    _close(ev.required_energy, z["trj_req_e"][t, p], "required_energy", 1e-9)
            _close(cs.current_power_output, z["trj_cs_power"][t, i], "cs power")
        _close(env.transformers[0].current_power, z["trj_tr_power"][t, 0], "tr power")
        _close(env.current_power_usage[t], z["trj_usage"][t], "usage")
    env.close()


@pytest.mark.parametrize("use_torch", [True, False], ids=["torch_tensors", "ctypes_buffers"])
@pytest.mark.parametrize("cfg,sf,rf", [("V2GProfitPlusLoads.yaml", "V2G_profit_max_loads", "ProfitMax_TrPenalty_UserIncentives"),
                                       ("PublicPST.yaml", "PublicPST", "SquaredTrackingErrorReward")])
def test_vec_env_from_yaml_matches_oracle(cfg, sf, rf, use_torch):
    from ev2gym_amd import _abi
    from ev2gym_amd.baselines.heuristics import RandomAgent
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.vec_env import EV2GymVec
    from oracle.oracle import Oracle
    env = EV2GymVec(config_file=os.path.join(CFG, cfg), num_envs=96, state_function=sf, reward_function=rf, seed=3,
                    auto_reset=True, use_torch=use_torch)
    obs, _ = env.reset()
    ora = Oracle(_window(env), _abi.REWARD_KINDS[rf], _abi.STATE_KINDS[sf])
    to_np = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else (x.to_host() if hasattr(x, "to_host") else np.asarray(x))  # noqa: E731
    _close(to_np(obs), ora.reset(), "reset obs")
    agent = RandomAgent(seed=5)
    E, P, T = env.num_envs, env.number_of_ports, env.simulation_length
    lo = -1.0 if env.v2g_enabled else 0.0
    for t in range(T):
        a = agent.get_action(env)
        a_host = host_uniform(E * P, 5 * 1000003 + t + 1, lo, 1.0).reshape(E, P)
        assert np.array_equal(to_np(a), a_host)
        obs, rew, done, trunc, info = env.step(a)
        o_obs, o_rew, o_done, o_mask, rc = ora.step(a_host)
        _close(to_np(rew), o_rew, f"reward[{t}]")
        assert np.array_equal(to_np(done), o_done)
        assert np.array_equal(to_np(info["action_mask"]), o_mask)
        if t < T - 1:
            _close(to_np(obs), o_obs, f"obs[{t}]")
    # auto_reset: the terminal step returns the reset observation, the terminal one rides in info, stats are per env
    _close(to_np(info["terminal_observation"]), o_obs, "terminal obs")
    st = ora.stats()
    ora2 = Oracle(_window(env), _abi.REWARD_KINDS[rf], _abi.STATE_KINDS[sf])   # the auto-reset drew fresh scenarios from the pool
    _close(to_np(obs), ora2.reset(), "obs after auto-reset")
    _close(info["total_profits"], st[:, 1], "total_profits")
    _close(info["total_reward"], st[:, 16], "total_reward")
    assert env.current_step == 0
    env.close()


def test_vec_env_rejects_unfused_plugins():
    from ev2gym_amd.vec_env import EV2GymVec
    with pytest.raises(NotImplementedError):
        EV2GymVec(config_file=os.path.join(CFG, "PublicPST.yaml"), num_envs=4, state_function=lambda env: None)


@pytest.mark.parametrize("use_torch,obs_dtype,copy_obs", [(False, np.float64, True), (True, np.float64, True), (False, np.float32, True), (False, np.float32, False)],
                         ids=["ctypes_buffers", "torch_tensors", "float32_fast_path", "float32_views_of_pinned_blocks"])
def test_sb3_vec_env_protocol_matches_oracle(use_torch, obs_dtype, copy_obs):
    """SB3 VecEnv protocol (step_async/step_wait, reset at episode end with terminal_observation, numpy out); with float32
    observations and engine-owned buffers the adapter hands float32 over in both directions (its fast path)."""
    from ev2gym_amd import _abi
    from ev2gym_amd.sb3_vec_env import EV2GymSB3VecEnv
    from oracle.oracle import Oracle
    sf, rf = "V2G_profit_max_loads", "ProfitMax_TrPenalty_UserIncentives"
    venv = EV2GymSB3VecEnv(config_file=os.path.join(CFG, "V2GProfitPlusLoads.yaml"), num_envs=24, state_function=sf,
                           reward_function=rf, seed=9, use_torch=use_torch, obs_dtype=obs_dtype, copy_obs=copy_obs)
    otol = 1e-9 if obs_dtype == np.float64 else 2e-7     # float32 observations: one rounding
    assert venv._fast == (obs_dtype == np.float32)
    E, P, T = venv.num_envs, venv.vec.number_of_ports, venv.vec.simulation_length
    assert venv.observation_space.shape == (venv.vec.obs_dim,) and venv.action_space.shape == (P,)
    assert venv.env_is_wrapped(object) == [False] * E and venv.get_attr("simulation_length", 0) == [T]
    obs = venv.reset()
    ora = Oracle(_window(venv.vec), _abi.REWARD_KINDS[rf], _abi.STATE_KINDS[sf])
    assert isinstance(obs, np.ndarray) and obs.shape == (E, venv.vec.obs_dim)
    _close(obs, ora.reset(), "reset obs", tol=otol)
    rng = np.random.default_rng(4)
    ret = np.zeros(E)
    for t in range(T + 3):            # runs across the episode boundary
        a = rng.uniform(-1, 1, (E, P)).astype(np.float32)     # SB3 hands float32 actions
        prev, prev_copy = obs, obs.copy()
        venv.step_async(a)
        obs, rew, done, infos = venv.step_wait()
        assert np.array_equal(prev, prev_copy) and (copy_obs or not np.shares_memory(prev, obs))   # the array of the step before is still intact (views: two blocks in turn)
        if t == T:
            ret[:] = 0.0
        o_obs, o_rew, o_done, o_mask, rc = ora.step(a.astype(np.float64))
        ret += o_rew
        assert rew.dtype == np.float32 and done.dtype == bool and len(infos) == E
        _close(rew.astype(np.float64), o_rew, f"reward[{t}]", tol=1e-6)   # float32 on the SB3 side
        assert np.array_equal(done, o_done.astype(bool))
        assert all(np.array_equal(infos[i]["action_mask"], o_mask[i]) for i in range(E))
        if done.all():
            st = ora.stats()
            _close(np.stack([i["terminal_observation"] for i in infos]), o_obs, "terminal obs", tol=otol)
            _close(np.array([i["total_profits"] for i in infos]), st[:, 1], "total_profits")
            _close(np.array([i["episode"]["r"] for i in infos]), ret, "episode return")
            assert all(i["episode"]["l"] == T and i["TimeLimit.truncated"] is False for i in infos)
            ora.close()   # the reset inside step_wait drew fresh scenarios from the pool
            ora = Oracle(_window(venv.vec), _abi.REWARD_KINDS[rf], _abi.STATE_KINDS[sf])
            _close(obs, ora.reset(), "obs after reset inside step_wait", tol=otol)
        else:
            assert "terminal_observation" not in infos[0]
            _close(obs, o_obs, f"obs[{t}]", tol=otol)
            assert obs.dtype == obs_dtype
    venv.close()


@pytest.mark.parametrize("name", ["replay_v2gppl_p2_rand_s21", "replay_pst_rand_s22"])
def test_facade_from_replay_file_reproduces_reference_episode(name, tmp_path):
    """EV2Gym(load_from_replay_path=...) on the engine == the reference env built from the same pickle."""
    from ev2gym_amd.env import EV2Gym
    from ev2gym_amd.vec_env import EV2GymVec
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    pkl = tmp_path / "replay_sim.pkl"
    pkl.write_bytes(bytes(z["replay_pkl"]))
    sf, rf = str(z["case"][2]), str(z["case"][3])
    cfg = os.path.join(CFG, "PublicPST.yaml") if str(z["case"][1]) == "PublicPST.yaml" else None   # optional: v2g flag only
    env = EV2Gym(config_file=cfg, load_from_replay_path=str(pkl), state_function=sf, reward_function=rf)
    assert env.action_space.low[0] == (0.0 if cfg else -1.0)
    obs, _ = env.reset()
    _close(obs, z["trj_obs"][0], "reset obs")
    for t in range(len(z["act"])):
        obs, rew, done, trunc, info = env.step(z["act"][t].copy())
        _close(obs, z["trj_obs"][t + 1], f"obs[{t}]")
        _close(rew, z["trj_reward"][t], f"reward[{t}]")
        assert (info["action_mask"] == z["trj_mask"][t]).all()
    for i, k in enumerate(__import__("ev2gym_amd")._abi.STAT_NAMES):
        _close(info[k], z["trj_stats"][i], k)
    assert info["voltage_violation"] == 0
    env.close()
    # two copies of the file as a 2-env batch
    venv = EV2GymVec(load_from_replay_path=[str(pkl), str(pkl)], state_function=sf, reward_function=rf, use_torch=False)
    assert venv.num_envs == 2
    o, _ = venv.reset()
    _close(o[1], z["trj_obs"][0], "vec reset obs")
    o, r, d, tr, inf = venv.step(np.stack([z["act"][0], z["act"][0]]))
    _close(o[0], z["trj_obs"][1], "vec obs[0]")
    _close(r[1], z["trj_reward"][0], "vec reward[0]")
    venv.close()


def test_save_replay_writes_a_file_the_facade_replays(tmp_path):
    """save_replay=True (ev2gym_env.py:474-475,503-510): the finished episode is written as replay_<sim_name>.pkl; an env built
    from that file re-runs the same scenario (same observations / rewards for the same actions) and carries the run's totals."""
    import glob
    from ev2gym_amd.env import EV2Gym
    from ev2gym_amd.replay import read_replay_object
    kw = dict(state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives")
    cfg = os.path.join(CFG, "V2GProfitPlusLoads.yaml")
    a = EV2Gym(config_file=cfg, seed=11, save_replay=True, replay_save_path=str(tmp_path) + "/", **kw)
    obs0, _ = a.reset(seed=11)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (a.simulation_length, a.number_of_ports))
    trj = [a.step(acts[t].copy()) for t in range(a.simulation_length)]
    files = glob.glob(str(tmp_path / "replay_*.pkl"))
    assert files == [str(tmp_path / f"replay_{a.sim_name}.pkl")]
    rep = read_replay_object(files[0])
    assert rep.stats["total_ev_served"] == trj[-1][4]["total_ev_served"]
    np.testing.assert_allclose(sum(c.total_profits for c in rep.charging_stations), trj[-1][4]["total_profits"], rtol=1e-12)
    np.testing.assert_allclose(rep.ev_load_potential, a.current_power_usage)
    b = EV2Gym(config_file=cfg, load_from_replay_path=files[0], **kw)
    assert b.sim_name == a.sim_name + "_replay"
    o, _ = b.reset()
    # the recorded episode overwrote the forecasts with actuals (transformer.py:178-180), so only forecast-free entries agree at t=0
    for t in range(a.simulation_length):
        o, r, d, _, info = b.step(acts[t].copy())
        _close(r, trj[t][1], f"reward[{t}]")
        assert (info["action_mask"] == trj[t][4]["action_mask"]).all()
    for k in ("total_ev_served", "total_profits", "total_energy_charged", "total_energy_discharged", "average_user_satisfaction"):
        _close(info[k], trj[-1][4][k], k)
    a.close(); b.close()


@pytest.mark.parametrize("name", ["plugin_pst_sqtr_rand_s23", "plugin_pst_surplus_rand_s24", "plugin_pst_idlepen_mixed_s25",
                                  "plugin_v2gppl_sqtr_rand_s26", "plugin_v2gppl_profitmax_rand_s29", "plugin_v2gppl_c10r3_sqtr_mixed_s31"])
def test_host_evaluated_rewards_through_the_facade(name):
    """Reward callables WITHOUT the fusion marker (user plugins; here the Python bodies of reference built-ins, marker stripped)
    are evaluated on the host: the facade feeds them the same env attributes the reference does, so the reference's reward
    trajectory is reproduced -- next to a host-evaluated cost function."""
    from ev2gym_amd.env import EV2Gym
    from ev2gym_amd.rl_agent import cost as C, reward as R, state as S
    from ev2gym_amd.scenario import ScenarioBatch
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    batch = ScenarioBatch.from_single(z)
    rf0 = getattr(R, str(z["case"][3]))
    rf = lambda env, *a: rf0(env, *a)  # noqa: E731
    env = EV2Gym(scenario=batch, state_function=getattr(S, str(z["case"][2])), reward_function=rf,
                 cost_function=C.transformer_overload_usrpenalty_cost)
    obs, _ = env.reset()
    _close(obs, z["trj_obs"][0], "reset obs")
    for t in range(len(z["act"])):
        obs, rew, done, trunc, info = env.step(z["act"][t].copy())
        _close(obs, z["trj_obs"][t + 1], f"obs[{t}]")
        _close(rew, z["trj_reward"][t], f"reward[{t}]")
        # cost.py:8-18 == what ProfitMax_TrPenalty_UserIncentives subtracts: overload + dissatisfaction penalties
        want_cost = 100 * z["trj_tr_overload"][t].sum() + sum(100 * np.exp(-10 * s) for s in z["trj_dep_score"][t][:z["trj_n_departed"][t]])
        _close(env.cost if done else info["cost"], want_cost, f"cost[{t}]")
    _close(info["total_reward"], z["trj_stats"][16], "total_reward")
    assert env._host_reward
    env.close()


def test_reset_with_a_seed_draws_that_seeds_scenarios():
    """The reference draws a scenario inside every reset (ev2gym_env.py:243-296).  Facade: reset(seed=s) == constructing with
    seed s; reset() moves on to a new scenario, reproducibly for equal constructor seeds.  EV2GymVec: reset(seed=s) selects a
    window of the resident pool, reproducibly for equal constructor seeds."""
    from ev2gym_amd.env import EV2Gym
    from ev2gym_amd.vec_env import EV2GymVec
    cfg = os.path.join(CFG, "V2GProfitPlusLoads.yaml")
    kw = dict(state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives")
    a = EV2Gym(config_file=cfg, seed=5, **kw)
    b = EV2Gym(config_file=cfg, seed=9, **kw)
    first_a = a._arr["charge_price"].copy()
    assert not np.array_equal(a._arr["ev_t_arr"], b._arr["ev_t_arr"]) or not np.array_equal(a._arr["charge_price"], b._arr["charge_price"])
    ob, _ = b.reset(seed=5)
    oa, _ = a.reset(seed=5)
    assert np.array_equal(oa, ob) and len(a.EVs_profiles) == len(b.EVs_profiles) and np.array_equal(a._arr["charge_price"], first_a)
    rng = np.random.default_rng(0)
    for t in range(40):
        act = rng.uniform(-1, 1, a.number_of_ports)
        ra, rb = a.step(act.copy()), b.step(act.copy())
        assert np.array_equal(ra[0], rb[0]) and ra[1] == rb[1]
    # reset() without a seed: a NEW scenario (the reference's behaviour), the same one for envs built with the same seed
    c = EV2Gym(config_file=cfg, seed=5, **kw)
    a.reset(seed=5)
    a.reset(); c.reset()
    assert not np.array_equal(a._arr["charge_price"], first_a) or len(a.EVs_profiles) != len(b.EVs_profiles)
    assert np.array_equal(a._arr["charge_price"], c._arr["charge_price"]) and np.array_equal(a._arr["ev_t_arr"], c._arr["ev_t_arr"])
    a.resample_on_reset = False     # opt out: reset() re-arms the current scenario
    cur = a._arr["charge_price"].copy()
    a.reset()
    assert np.array_equal(a._arr["charge_price"], cur)
    a.close(); b.close(); c.close()
    v1 = EV2GymVec(config_file=cfg, num_envs=16, seed=3, use_torch=False, **kw)
    v2 = EV2GymVec(config_file=cfg, num_envs=16, seed=3, use_torch=False, **kw)
    o2, _ = v2.reset(seed=11)
    o1, _ = v1.reset(seed=11)
    assert np.array_equal(o1, o2) and v1.engine.scenario_offset == v2.engine.scenario_offset
    act = rng.uniform(-1, 1, (16, v1.number_of_ports))
    s1, s2 = v1.step(act), v2.step(act)
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])
    offs = {v1.reset(seed=k) and v1.engine.scenario_offset for k in range(12)}
    assert len(offs) >= 4 and all(o % 16 == 0 for o in offs), "different seeds select different windows of the pool (8 aligned windows of 16 envs)"
    v1.close(); v2.close()

