"""Randomised sweep of the engine against the CPU oracle: shapes, plugins, timescales, topology files, action policies, launch
modes and pool windows drawn from a fixed seed -- every kernel the routing can pick, in combinations no hand-written test names."""
import numpy as np
import pytest

import os

pytestmark = pytest.mark.gpu
N_CASES = 120
SEED_OFFSET = int(os.environ.get("EV2G_FUZZ_OFFSET", "0"))   # EV2G_FUZZ_OFFSET=1000 pytest ...: the same sweep over other draws


def _close(a, b, what, tol=1e-9):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert (np.isnan(a) == np.isnan(b)).all(), f"{what}: NaN pattern differs"
    err = np.nan_to_num(np.abs(a - b) / np.maximum(1.0, np.abs(np.nan_to_num(b))))
    assert err.max(initial=0.0) <= tol, f"{what}: max rel err {err.max():.3e}"


def _close_reward(rk, r_eng, r_ora, ora, t, what):
    """Rewards to 1e-9, INCLUDING the one discontinuity among the reference's reward functions: SquaredTrackingErrorRewardWithPenalty
    (kind 4, reward.py:46-58) subtracts 100 when `current_power_usage == 0`, an exact test on a sum.  With V2G, charging and discharging
    powers that cancel leave a rounding residue (~1e-15) or an exact zero depending on the ORDER of the sum; the reference adds charger by
    charger.  The kernels' fixed summation tree used to differ from it by exactly that 100 in such steps (rounds 1-2 tolerated it here);
    they now repeat the sum in the reference's order for this reward (RewardIn::usage_seq, csrc/ev2g_device.h)."""
    _close(np.asarray(r_eng, float), np.asarray(r_ora, float), what)


def _draw(case):
    case = case + SEED_OFFSET
    from ev2gym_amd.scenario_gen import GenConfig
    rng = np.random.default_rng(9000 + case)
    v2g = bool(rng.random() < 0.7)
    kw = dict(n_envs=int(rng.integers(3, 40)), number_of_charging_stations=int(rng.integers(2, 45)),
              number_of_ports_per_cs=int(rng.choice([1, 1, 1, 2, 3])), number_of_transformers=int(rng.integers(1, 5)),
              timescale=int(rng.choice([5, 15, 15, 30])), scenario=str(rng.choice(["workplace", "public", "private"])),
              simulation_days=str(rng.choice(["weekdays", "weekends", "both"])), v2g_enabled=v2g,
              heterogeneous_ev_specs=bool(rng.random() < 0.7), fleet_with_efficiency_tables=bool(rng.random() < 0.6),
              power_setpoint_enabled=bool(rng.random() < 0.5), inflexible_loads=bool(rng.random() < 0.7), solar_power=bool(rng.random() < 0.7),
              demand_response=bool(rng.random() < 0.6), dr_events_per_day=int(rng.integers(1, 3)), spawn_multiplier=float(rng.choice([3, 5, 10])),
              seed=case)
    if not v2g:
        kw.update(cs_max_discharge_current=0.0)
    if kw["timescale"] == 5:
        kw["simulation_length"] = 96
    if rng.random() < 0.25:   # a topology file's chargers (falling port counts: the reference's mask index stays inside the array)
        nps = np.array(sorted(rng.integers(1, 5, int(rng.integers(3, 12))).tolist(), reverse=True))
        C = len(nps)
        R = int(rng.integers(1, min(4, C) + 1))
        kw["topology"] = dict(n_ports=nps, transformer=np.sort(rng.integers(0, R, C)) if R > 1 else np.zeros(C, int),
                              min_charge_current=np.full(C, float(rng.choice([0, 6]))), max_charge_current=rng.choice([16.0, 32.0], C),
                              min_discharge_current=np.zeros(C), max_discharge_current=(rng.choice([-16.0, -32.0], C) if v2g else np.zeros(C)),
                              voltage=rng.choice([230.0, 400.0], C), phases=rng.choice([1, 3], C), tr_max_power=rng.choice([40.0, 60.0, 100.0], R))
        kw["topology"]["transformer"] = np.unique(kw["topology"]["transformer"], return_inverse=True)[1]   # contiguous ids
        kw["topology"]["tr_max_power"] = kw["topology"]["tr_max_power"][:kw["topology"]["transformer"].max() + 1]
    return rng, GenConfig(**kw)


def _back_to_back(pool):
    """Extends every stay up to the step before the same port's next arrival: the next EV plugs in at the end of the very step
    its predecessor leaves in (ev2gym_env.py:363-417 -- departures precede the spawns of a step).  The reference's spawner keeps a gap
    between sessions of a port, a replayed scenario need not."""
    from ev2gym_amd.scenario import resolve_ports
    a, port, st = pool.arrays, resolve_ports(pool), pool.arrays["env_session_start"]
    for e in range(pool.n_envs):
        last = {}
        for s in range(st[e], st[e + 1]):
            if port[s] in last:
                a["ev_t_dep"][last[port[s]]] = a["ev_t_arr"][s] - 1
            last[port[s]] = s
    assert np.array_equal(resolve_ports(pool), port)


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_configuration_matches_oracle(case):
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, EngineError, host_uniform
    from ev2gym_amd.scenario_gen import generate
    from oracle.oracle import Oracle
    rng, cfg = _draw(case)
    pool = generate(cfg)
    if case % 3 == 1:
        _back_to_back(pool)
    M = pool.n_envs
    E = int(rng.integers(max(1, M // 2), M + 1))
    rk, sk = int(rng.integers(0, 11)), int(rng.integers(0, 3))
    cost_kind = int(rng.choice([0, 1, 2])) if rk not in (3, 8, 9, 10) else int(rng.choice([0, 2]))
    flags = 4 | (1 if rng.random() < 0.3 else 0)
    eng = Engine(pool, rk, sk, device=0, flags=flags, cost_kind=cost_kind, n_active_envs=E)
    P, D, T = eng.P, eng.D, eng.T
    lo = -1.0 if cfg.v2g_enabled else 0.0
    pol = str(rng.choice(["rand", "wild", "sparse"]))
    acts = host_uniform(T * E * P, 100 + case, lo * (1.5 if pol == "wild" else 1.0), 1.5 if pol == "wild" else 1.0).reshape(T, E, P)
    if pol == "sparse":
        acts = acts * (np.random.default_rng(case).random((T, E, P)) < 0.6)
    d_act = eng.empty((T, E, P)).upload(acts)
    d_obs, d_rew, d_mask, d_done = eng.empty((T, E, D)), eng.empty((T, E)), eng.empty((T, E, P), np.uint8), eng.empty((T, E), np.uint8)
    d_cost = eng.empty((T, E)) if cost_kind else None
    if d_cost is not None:
        eng.set_extras(cost=d_cost, cost_stride=E)
    off = int(rng.integers(0, 3 * M))
    ora = Oracle(pool.select((np.arange(E) + off) % M), rk, sk)
    d_obs0 = eng.empty((E, D))
    eng.reset(d_obs0, offset=off)
    _close(d_obs0.to_host(), ora.reset(), "reset obs")
    k1 = int(rng.integers(0, T))            # a persistent launch, then single-step launches, then a persistent one to the end
    k2 = int(rng.integers(k1, min(T, k1 + 12) + 1))
    if k1:
        eng.step_n(k1, d_act, E * P, d_obs, E * D, d_rew, E, d_done, E, d_mask, E * P, auto_reset=False, persistent=True)
    for t in range(k1, k2):
        if d_cost is not None:
            eng.set_extras(cost=d_cost.at(t * E), cost_stride=E)
        eng.step(d_act.at(t * E * P), d_obs.at(t * E * D), d_rew.at(t * E), d_done.at(t * E), d_mask.at(t * E * P))
    if k2 < T:
        if d_cost is not None:
            eng.set_extras(cost=d_cost.at(k2 * E), cost_stride=E)
        eng.step_n(T - k2, d_act.at(k2 * E * P), E * P, d_obs.at(k2 * E * D), E * D, d_rew.at(k2 * E), E, d_done.at(k2 * E), E,
                   d_mask.at(k2 * E * P), E * P, auto_reset=False, persistent=True)
    obs, rew, mask, done = d_obs.to_host(), d_rew.to_host(), d_mask.to_host(), d_done.to_host()
    cost = d_cost.to_host() if d_cost is not None else None
    faulted = False
    tag = f"case {case} {eng.kernel_name} rk={rk} sk={sk} P={P} R={eng.R} dt={pool.timescale} {pol}"
    for t in range(T):
        o, r, d, m, rc = ora.step(acts[t].copy())
        faulted = faulted or rc != 0
        assert np.array_equal(mask[t], m), f"{tag}: mask[{t}]"
        _close(obs[t], o, f"{tag}: obs[{t}]")
        _close_reward(rk, rew[t], r, ora, t, f"{tag}: reward[{t}]")
        assert np.array_equal(done[t], d), f"{tag}: done[{t}]"
    _close(eng.stats(), ora.stats(), f"{tag}: statistics")
    if faulted:
        with pytest.raises(EngineError, match="over-current"):
            eng.check_faults()
    else:
        eng.check_faults()
    eng.close()
    ora.close()


@pytest.mark.parametrize("case", range(40))
def test_random_fused_runs_across_episode_ends(case):
    """ev2g_step_n across one or two episode ends with in-run resets (SAME scenarios / NEXT window of the pool), persistent or
    step by step, float64 or float32 actions, every kernel the drawn shape routes to -- against oracle episodes on the windows the
    engine reports."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, host_uniform
    from ev2gym_amd.scenario_gen import generate
    from oracle.oracle import Oracle
    rng, cfg = _draw(500 + case)
    cfg.n_envs = int(rng.integers(6, 30))
    if rng.random() < 0.15:     # a big env now and then: the 512 / 1024-thread general kernel and the generic one
        cfg.topology = None
        cfg.number_of_charging_stations, cfg.number_of_ports_per_cs, cfg.n_envs = int(rng.choice([300, 700, 1100])), 1, int(rng.integers(3, 6))
        cfg.number_of_transformers = int(rng.choice([1, 7, 50]))
    pool = generate(cfg)
    M = pool.n_envs
    E = int(rng.integers(max(1, M // 3), M + 1))
    rk, sk = int(rng.integers(0, 11)), int(rng.integers(0, 3))
    eng = Engine(pool, rk, sk, device=0, flags=4, n_active_envs=E)
    P, D, T = eng.P, eng.D, eng.T
    mode = _abi.AUTO_RESET_NEXT if rng.random() < 0.6 else _abi.AUTO_RESET_SAME
    persistent = bool(rng.random() < 0.6)
    f32 = bool(rng.random() < 0.4)
    K = T + int(rng.integers(1, T + 6))
    lo = -1.0 if cfg.v2g_enabled else 0.0
    acts = host_uniform(K * E * P, 700 + case, lo, 1.0).reshape(K, E, P)
    if f32:
        acts = acts.astype(np.float32).astype(np.float64)
        d_act32 = eng.empty((K, E, P), np.float32).upload(acts.astype(np.float32))
        eng.set_extras(actions_f32=d_act32)
    d_act = None if f32 else eng.empty((K, E, P)).upload(acts)
    d_obs, d_rew, d_done, d_mask = eng.empty((K, E, D)), eng.empty((K, E)), eng.empty((K, E), np.uint8), eng.empty((K, E, P), np.uint8)
    off0 = int(rng.integers(0, M))
    eng.reset(offset=off0)
    eng.step_n(K, d_act, E * P, d_obs, E * D, d_rew, E, d_done, E, d_mask, E * P, auto_reset=mode, persistent=persistent)
    obs, rew, done, mask = d_obs.to_host(), d_rew.to_host(), d_done.to_host(), d_mask.to_host()
    tag = f"case {case} {eng.kernel_name} rk={rk} sk={sk} P={P} R={eng.R} mode={mode} persistent={persistent} f32={f32}"
    adv = (E % M) if mode == _abi.AUTO_RESET_NEXT else 0
    n_ep = (K - 1) // T
    assert eng.scenario_offset == (off0 + n_ep * adv) % M and eng.current_step == K - n_ep * T, tag
    k = 0
    for ep in range(n_ep + 1):
        ora = Oracle(pool.select((np.arange(E) + off0 + ep * adv) % M), rk, sk)
        ora.reset()
        for t in range(min(T, K - k)):
            o, r, d, m, rc = ora.step(acts[k].copy())
            assert np.array_equal(mask[k], m), f"{tag}: mask[{k}]"
            _close(obs[k], o, f"{tag}: obs[{k}]")
            _close_reward(rk, rew[k], r, ora, t, f"{tag}: reward[{k}]")
            assert np.array_equal(done[k], d), f"{tag}: done[{k}]"
            k += 1
        if ep == n_ep:   # the running episode: per-env state through the inspection API
            for e in (0, E - 1):
                pk, po = eng.peek(e), ora.peek(e)
                _close(pk["port_capacity"], po["cap"], f"{tag}: capacity")
                assert (pk["port_session"] == po["session"]).all(), tag
            se, so = eng.stats(), ora.stats()
            worst = {}
            for i, name in enumerate(_abi.STAT_NAMES):
                x, y = se[:, i], so[:, i]
                err = np.nan_to_num(np.abs(x - y) / np.maximum(1.0, np.abs(np.nan_to_num(y))))
                if err.max() > 1e-9 or (np.isnan(x) != np.isnan(y)).any():
                    j = int(err.argmax())
                    worst[name] = (float(x[j]), float(y[j]), j)
            assert not worst, f"{tag}: statistics of the running episode (step {eng.current_step}): {worst}"
        ora.close()
    eng.close()
