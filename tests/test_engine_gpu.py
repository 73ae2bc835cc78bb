"""Parity of the HIP engine (through the C-ABI) against the reference goldens and the CPU oracle.

Bar (BASELINE.json north_star): <=1e-6 relative on float64 SoC / power, bit-exact on EV arrival /
departure indexing.  We hold the engine to a much tighter 1e-9: per-port arithmetic follows the
reference's operation order (-ffp-contract=off) and is expected to be bit-identical up to device
`exp`; only the per-transformer / per-env sums use a different (fixed-tree) order.
"""
import numpy as np
import pytest

from conftest import GOLDEN_FILES, GOLDEN_IDS, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _close(a, b, what, tol=RTOL):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert (np.isnan(a) == np.isnan(b)).all(), f"{what}: NaN pattern (port occupancy) differs"
    scale = np.maximum(1.0, np.abs(np.nan_to_num(b)))
    err = np.nan_to_num(np.abs(a - b) / scale)
    assert err.max(initial=0.0) <= tol, f"{what}: max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


def _engine(batch, rk, sk, flags=1 | 4, **kw):
    from ev2gym_amd.engine import Engine
    return Engine(batch, rk, sk, device=0, flags=flags, **kw)


def _expected_kernel(batch, sk, rk, forced_v2=False):
    """The routing rule of ev2g_load_scenarios, restated: the common shape gets the fast path, with or without the
    charger-history flag."""
    P, R, npc = batch.n_ports, batch.n_transformers, batch.ports_per_charger
    if not batch.uniform_ports:     # topology file with different port counts per charger
        return "ev2g_step_kernel"
    if 2 <= P <= 64 and R == 1 and npc == 1 and not forced_v2:
        return f"ev2g_step_wave<{sk},{min(rk, 3)}>"
    return f"ev2g_step_v2<{256 if P <= 256 else 512 if P <= 512 else 1024}>" if P <= 1024 else "ev2g_step_kernel"


@pytest.mark.parametrize("flags", [4, 1 | 4], ids=["soclog", "cshist+soclog"])
@pytest.mark.parametrize("path", GOLDEN_FILES, ids=GOLDEN_IDS)
def test_engine_matches_reference_golden(path, flags):
    """Every reference fixture through the kernel the engine routes its shape to -- for the single-port, one-transformer
    fixtures that is ev2g_step_wave, the benchmarked kernel, with and without the charger-history flag."""
    z, batch, rk, sk = load_golden(path)
    eng = _engine(batch, rk, sk, flags=flags)
    assert eng.kernel_name == _expected_kernel(batch, sk, rk), (eng.kernel_name, eng.fallback_reason)
    log_cs = bool(flags & 1)
    E, P, D = eng.E, eng.P, eng.D
    assert D == z["trj_obs"].shape[1]
    d_act, d_obs = eng.empty((E, P)), eng.empty((E, D))
    d_rew, d_done, d_mask = eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    eng.reset(d_obs)
    _close(d_obs.to_host()[0], z["trj_obs"][0], "reset obs")
    nT = len(z["act"])
    for t in range(nT):
        d_act.upload(z["act"][t:t + 1])
        eng.step(d_act, d_obs, d_rew, d_done, d_mask)
        _close(d_obs.to_host()[0], z["trj_obs"][t + 1], f"obs[{t}]")
        _close(d_rew.to_host()[0], z["trj_reward"][t], f"reward[{t}]")
        assert d_done.to_host()[0] == z["trj_done"][t]
        assert (d_mask.to_host()[0] == z["trj_mask"][t]).all(), f"action_mask[{t}] (arrival/departure indexing)"
        pk = eng.peek(0)
        _close(pk["port_capacity"], z["trj_cap"][t], f"capacity[{t}]")
        _close(pk["port_energy"], z["trj_energy"][t], f"energy[{t}]")
        _close(pk["port_current"], z["trj_current"][t], f"current[{t}]")
        _close(pk["port_total_energy"], z["trj_tot_e"][t], f"tot_e[{t}]")
        _close(pk["port_prev_power"], z["trj_prev_power"][t], f"prev_power[{t}]")
        assert (pk["port_cycles"] == z["trj_cycles"][t]).all(), f"cycles[{t}]"
        if log_cs:
            _close(pk["cs_power"], z["trj_cs_power"][t], f"cs_power[{t}]")
            _close(pk["cs_amps"], z["trj_cs_amps"][t], f"cs_amps[{t}]")
            _close(pk["cs_profits"], z["trj_cs_profits"][t], f"cs_profits[{t}]")
            _close(pk["cs_energy_charged"], z["trj_cs_e_ch"][t], f"cs_e_ch[{t}]")
            _close(pk["cs_energy_discharged"], z["trj_cs_e_dis"][t], f"cs_e_dis[{t}]")
        _close(pk["tr_power"], z["trj_tr_power"][t], f"tr_power[{t}]")
    pk = eng.peek(0)
    _close(pk["power_usage"][:nT], z["trj_usage"], "current_power_usage")
    _close(pk["power_potential"][:nT], z["trj_potential"], "charge_power_potential")
    _close(pk["tr_overload"][:, :nT].T, z["trj_tr_overload"], "tr_overload")
    m = z["trj_ev_port"] >= 0
    assert (pk["session_port"][m] == z["trj_ev_port"][m]).all(), "first-free port assignment"
    _close(pk["session_afap"][m], z["trj_ev_afap"][m], "max_energy_AFAP")
    eng.check_faults()
    if "trj_stats" in z:
        from ev2gym_amd import _abi
        st = eng.stats()[0]
        for i, k in enumerate(_abi.STAT_NAMES):
            g = z["trj_stats"][i]
            if np.isnan(g):
                assert np.isnan(st[i]), k
            else:
                assert abs(st[i] - g) <= 1e-9 * max(1.0, abs(g)), (k, st[i], g)
        with pytest.raises(Exception):
            eng.step(d_act, d_obs, d_rew, d_done, d_mask)  # `assert not self.done` ev2gym_env.py:343
    eng.close()


def _tiled(name, E):
    import os
    from conftest import GOLDEN_DIR
    z, batch, rk, sk = load_golden(os.path.join(GOLDEN_DIR, name + ".npz"))
    return batch.tile(E), rk, sk


@pytest.mark.parametrize("name,E,lo", [("v2gppl_c50_rand_s9", 333, -1.0), ("pst_rand_s2", 517, 0.0),
                                       ("v2gppl_c60r5_rand_s13", 97, -1.0), ("v2gppl_p2_rand_s11", 130, -1.3),
                                       ("pst_p3_rand_s12", 77, 0.0), ("v2gppl_c1000r50_rand_s15", 9, -1.0),
                                       ("v2gppl_ts5_rand_s16", 64, -1.0), ("pst_ts30_rand_s17", 65, 0.0)])   # 60/dt = 12, 2
def test_engine_matches_oracle_batched(name, E, lo):
    """Many envs per launch, a different action stream per env: engine == CPU oracle at every step."""
    from ev2gym_amd.engine import host_uniform
    from oracle.oracle import Oracle
    batch, rk, sk = _tiled(name, E)
    eng = _engine(batch, rk, sk, flags=4)
    ora = Oracle(batch, rk, sk)
    P, D, T = eng.P, eng.D, eng.T
    d_act, d_obs = eng.empty((E, P)), eng.empty((E, D))
    d_rew, d_done, d_mask = eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    eng.reset(d_obs)
    _close(d_obs.to_host(), ora.reset(), "reset obs")
    nT = T if P < 500 else 24
    for t in range(nT):
        eng.fill_uniform(d_act, E * P, 77 + t, lo, 1.0)
        a = host_uniform(E * P, 77 + t, lo, 1.0).reshape(E, P)
        assert (d_act.to_host() == a).all(), "device and host action generators must agree bit for bit"
        eng.step(d_act, d_obs, d_rew, d_done, d_mask)
        obs, rew, done, mask, rc = ora.step(a)
        assert rc == 0
        assert (d_mask.to_host() == mask).all(), f"mask[{t}]"
        _close(d_obs.to_host(), obs, f"obs[{t}]")
        _close(d_rew.to_host(), rew, f"reward[{t}]")
        assert (d_done.to_host() == done).all()
    for e in (0, E // 2, E - 1):
        pk, po = eng.peek(e), ora.peek(e)
        _close(pk["port_capacity"], po["cap"], "capacity")
        _close(pk["port_total_energy"], po["tot_e"], "tot_e")
        assert (pk["port_cycles"] == po["cycles"]).all()
        assert (pk["port_session"] == po["session"]).all()
        _close(pk["power_usage"], po["usage"], "usage")
        _close(pk["power_potential"], po["potential"], "potential")
        _close(pk["tr_overload"], po["tr_overload"], "tr_overload")
    if nT == T:
        st, so = eng.stats(), ora.stats()
        _close(st, so, "episode stats (incl. battery degradation from the SoC log)")
    eng.check_faults()
    eng.close()
    ora.close()


@pytest.mark.parametrize("name,E", [("v2gppl_c50_rand_s9", 260), ("pst_rand_s3", 100)])
def test_persistent_multi_step_launch_is_bit_identical(name, E):
    """ev2g_step_n: one persistent launch for K steps == K single-step launches, across an episode boundary."""
    batch, rk, sk = _tiled(name, E)
    eng = _engine(batch, rk, sk, flags=0)
    P, D, T = eng.P, eng.D, eng.T
    K = T + 9  # crosses the auto-reset
    lo = -1.0 if batch.v2g_enabled else 0.0
    d_act = eng.empty((K, E, P))
    eng.fill_uniform(d_act, K * E * P, 5, lo, 1.0)
    outs = []
    for persistent in (False, True):
        d_obs, d_rew = eng.empty((K, E, D)), eng.empty((K, E))
        d_done, d_mask = eng.empty((K, E), np.uint8), eng.empty((K, E, P), np.uint8)
        eng.reset()
        eng.step_n(K, d_act, E * P, d_obs, E * D, d_rew, E, d_done, E, d_mask, E * P, auto_reset=True,
                   persistent=persistent)
        assert eng.current_step == 9
        assert eng.last_step_n_kernel_ms() > 0
        outs.append((d_obs.to_host(), d_rew.to_host(), d_done.to_host(), d_mask.to_host()))
        for b in (d_obs, d_rew, d_done, d_mask):
            b.free()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert outs[0][2][T - 1].all() and not outs[0][2][T].any()
    # the second episode (after the in-kernel reset) replays the first one's scenario: same masks
    assert np.array_equal(outs[0][3][:9], outs[0][3][T:T + 9])
    eng.close()


def test_engine_is_deterministic():
    batch, rk, sk = _tiled("v2gppl_c60r5_rand_s13", 64)
    res = []
    for _ in range(2):
        eng = _engine(batch, rk, sk, flags=0)
        K, E, P, D = 40, eng.E, eng.P, eng.D
        d_act, d_obs, d_rew = eng.empty((K, E, P)), eng.empty((K, E, D)), eng.empty((K, E))
        eng.fill_uniform(d_act, K * E * P, 11, -1.0, 1.0)
        eng.step_n(K, d_act, E * P, d_obs, E * D, d_rew, E)
        res.append((d_obs.to_host(), d_rew.to_host()))
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("kernel", ["wave", "v2"])
@pytest.mark.parametrize("name,E,lo", [("v2gppl_c50_rand_s9", 131, -1.0), ("pst_rand_s2", 203, 0.0)])
def test_every_kernel_variant_matches_oracle(kernel, name, E, lo, monkeypatch):
    """The common shape (P <= 64, one transformer, single-port chargers) can run on two kernels: the wave-aligned default
    and (EV2G_KERNEL=v2) the general one.  Both must reproduce the oracle, persistent launch included."""
    from ev2gym_amd.engine import host_uniform
    from oracle.oracle import Oracle
    monkeypatch.setenv("EV2G_KERNEL", kernel)
    batch, rk, sk = _tiled(name, E)
    eng = _engine(batch, rk, sk, flags=4)
    assert eng.kernel_name == _expected_kernel(batch, sk, rk, forced_v2=(kernel == "v2"))
    ora = Oracle(batch, rk, sk)
    P, D, T = eng.P, eng.D, eng.T
    K = T
    d_act = eng.empty((K, E, P))
    eng.fill_uniform(d_act, K * E * P, 31, lo, 1.0)
    acts = host_uniform(K * E * P, 31, lo, 1.0).reshape(K, E, P)
    d_obs, d_rew, d_mask = eng.empty((K, E, D)), eng.empty((K, E)), eng.empty((K, E, P), np.uint8)
    eng.reset()
    eng.step_n(40, d_act, E * P, d_obs, E * D, d_rew, E, None, 0, d_mask, E * P, auto_reset=False, persistent=True)
    for k in range(40, K):   # finish the episode with single-step launches
        eng.step(d_act.at(k * E * P), d_obs.at(k * E * D), d_rew.at(k * E), None, d_mask.at(k * E * P))
    obs, rew, mask = d_obs.to_host(), d_rew.to_host(), d_mask.to_host()
    ora.reset()
    for k in range(K):
        o_obs, o_rew, o_done, o_mask, rc = ora.step(acts[k].copy())
        assert (mask[k] == o_mask).all(), f"mask[{k}]"
        _close(obs[k], o_obs, f"obs[{k}]")
        _close(rew[k], o_rew, f"reward[{k}]")
    _close(eng.stats(), ora.stats(), "episode stats")
    eng.check_faults()
    eng.close()
    ora.close()


@pytest.mark.parametrize("workload,K", [("cfg2", 112), ("cfg3", 112), ("cfg4", 6)])
def test_full_size_batch_properties_and_sampled_parity(workload, K):
    """BASELINE.json's full sizes (4096 x 50 / 8192 x 20, generated scenarios): what no small case shows -- every
    workgroup of a full grid, the XCD-aware group mapping, 32-bit offsets near their largest values.
      * a random sample of envs, replayed by the CPU oracle with the same action streams, matches at every step;
      * one persistent 112-step launch == 112 single-step launches, bit for bit (sampled envs + all rewards);
      * conservation: the per-env energy / profit totals of get_statistics equal the sums of the per-step port
        quantities the oracle accumulates for the sampled envs (already part of the oracle's statistics)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import WORKLOADS
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import generate
    from oracle.oracle import Oracle
    wl = WORKLOADS[workload]
    E = wl["envs"]
    batch = generate(wl["gen"](E, 7))
    rk, sk = _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]]
    eng = _engine(batch, rk, sk, flags=4)
    P, D, T = eng.P, eng.D, eng.T
    rng = np.random.default_rng(3)
    sample = np.sort(np.concatenate([[0, 1, E // 2, E - 2, E - 1], rng.choice(E, 27, replace=False)]))
    sample = np.unique(sample)
    K = min(K, T)   # cfg4: a few steps only (its [T,E,D] observation block would be 7 GB)
    d_act = eng.empty((K, E, P))
    eng.fill_uniform(d_act, K * E * P, 123, wl["lo"], 1.0)
    acts = host_uniform(K * E * P, 123, wl["lo"], 1.0).reshape(K, E, P)[:, sample]
    outs = []
    for persistent in (True, False):
        d_obs, d_rew, d_mask = eng.empty((K, E, D)), eng.empty((K, E)), eng.empty((K, E, P), np.uint8)
        eng.reset()
        eng.step_n(K, d_act, E * P, d_obs, E * D, d_rew, E, None, 0, d_mask, E * P, auto_reset=False, persistent=persistent)
        eng.check_faults()
        outs.append((d_obs.to_host()[:, sample], d_rew.to_host(), d_mask.to_host()[:, sample], eng.stats()))
        for b in (d_obs, d_rew, d_mask):
            b.free()
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert np.array_equal(a, b), "persistent launch != single-step launches at full size"
    assert np.array_equal(np.nan_to_num(outs[0][3]), np.nan_to_num(outs[1][3]))
    obs, rew, mask, stats = outs[0]
    ora = Oracle(batch.select(sample), rk, sk)
    ora.reset()
    for t in range(K):
        o, r, d, m, rc = ora.step(acts[t].copy())
        assert rc == 0
        assert np.array_equal(mask[t], m), f"mask[{t}]"
        _close(obs[t], o, f"obs[{t}]")
        _close(rew[t, sample], r, f"reward[{t}]")
    if K == T:
        _close(stats[sample], ora.stats(), "episode statistics of the sampled envs")
    ora.close()
    eng.close()


@pytest.mark.parametrize("P", [2, 3, 5, 8, 13, 16, 21, 31, 32, 33, 47, 64])
@pytest.mark.parametrize("kind", ["v2gppl", "pst", "v2gmax"])
def test_fast_path_for_every_env_width(P, kind):
    """The wave-aligned kernel packs 64 // P envs into a wavefront and splits the observation head, the history
    stores and the reduction over the env's lanes: every width class (1, 2, 3, 4, ... 32 envs per wavefront; fewer lanes
    than head-column pairs; two-port envs) against the oracle, a whole episode, persistent and single-step launches."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate
    from oracle.oracle import Oracle
    E = 37 if P > 8 else 150
    if kind == "v2gppl":
        batch = generate(GenConfig.v2g_profit_plus_loads(E, P, 1, seed=100 + P))
        rk, sk, lo = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"], -1.0
    elif kind == "v2gmax":   # the third fused pair (20-column observation head, no transformer penalty)
        batch = generate(GenConfig.v2g_profit_plus_loads(E, P, 1, seed=300 + P))
        rk, sk, lo = _abi.REWARD_KINDS["profit_maximization"], _abi.STATE_KINDS["V2G_profit_max"], -1.0
    else:
        batch = generate(GenConfig.public_pst(E, P, seed=200 + P))
        rk, sk, lo = _abi.REWARD_KINDS["SquaredTrackingErrorReward"], _abi.STATE_KINDS["PublicPST"], 0.0
    eng = _engine(batch, rk, sk, flags=4)
    ora = Oracle(batch, rk, sk)
    D, T = eng.D, eng.T
    d_act = eng.empty((T, E, P))
    eng.fill_uniform(d_act, T * E * P, 9, lo, 1.0)
    acts = host_uniform(T * E * P, 9, lo, 1.0).reshape(T, E, P)
    outs = []
    for persistent in (True, False):
        d_obs, d_rew, d_mask = eng.empty((T, E, D)), eng.empty((T, E)), eng.empty((T, E, P), np.uint8)
        eng.reset()
        eng.step_n(T, d_act, E * P, d_obs, E * D, d_rew, E, None, 0, d_mask, E * P, auto_reset=False, persistent=persistent)
        eng.check_faults()
        outs.append((d_obs.to_host(), d_rew.to_host(), d_mask.to_host(), eng.stats()))
        for b in (d_obs, d_rew, d_mask):
            b.free()
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert np.array_equal(a, b)
    obs, rew, mask, stats = outs[0]
    ora.reset()
    for t in range(T):
        o, r, d, m, rc = ora.step(acts[t].copy())
        assert rc == 0 and np.array_equal(mask[t], m), f"mask[{t}]"
        _close(obs[t], o, f"obs[{t}]")
        _close(rew[t], r, f"reward[{t}]")
    _close(stats, ora.stats(), "episode statistics")
    for e in (0, E - 1):
        pk, po = eng.peek(e), ora.peek(e)
        _close(pk["power_usage"], po["usage"], "usage history")
        _close(pk["power_potential"], po["potential"], "potential history")
        _close(pk["tr_overload"], po["tr_overload"], "overload history")
    eng.close()
    ora.close()


@pytest.mark.parametrize("C,npc,R", [(6, 2, 1), (7, 3, 2), (30, 1, 4), (130, 2, 5), (520, 1, 7), (1100, 1, 11), (700, 2, 50)])
def test_general_kernels_for_multi_port_and_multi_transformer_shapes(C, npc, R):
    """Shapes outside the fast path: multi-port chargers (action normalisation, first-free ports), several transformers
    (segmented reduction, round-robin charger map), 256 / 512 / 1024-thread workgroups of ev2g_step_v2 and, above 1024
    ports per env, the generic ev2g_step_kernel -- generated scenarios against the oracle, persistent == single-step."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate
    from oracle.oracle import Oracle
    E = 9 if C * npc <= 300 else 4
    batch = generate(GenConfig.v2g_profit_plus_loads(E, C, R, seed=C + R, number_of_ports_per_cs=npc))
    rk, sk = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"]
    eng = _engine(batch, rk, sk, flags=4)
    ora = Oracle(batch, rk, sk)
    P, D, T = eng.P, eng.D, eng.T
    assert P == C * npc
    K = T if P <= 300 else 40
    d_act = eng.empty((K, E, P))
    eng.fill_uniform(d_act, K * E * P, 17, -1.4, 1.4)      # beyond the action box: normalisation / clamp paths
    acts = host_uniform(K * E * P, 17, -1.4, 1.4).reshape(K, E, P)
    outs = []
    for persistent in (True, False):
        d_obs, d_rew, d_mask = eng.empty((K, E, D)), eng.empty((K, E)), eng.empty((K, E, P), np.uint8)
        eng.reset()
        eng.step_n(K, d_act, E * P, d_obs, E * D, d_rew, E, None, 0, d_mask, E * P, auto_reset=False, persistent=persistent)
        outs.append((d_obs.to_host(), d_rew.to_host(), d_mask.to_host()))
        for b in (d_obs, d_rew, d_mask):
            b.free()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    obs, rew, mask = outs[0]
    ora.reset()
    for t in range(K):
        o, r, d, m, rc = ora.step(acts[t].copy())
        assert np.array_equal(mask[t], m), f"mask[{t}]"
        _close(obs[t], o, f"obs[{t}]")
        _close(rew[t], r, f"reward[{t}]")
    if K == T:
        _close(eng.stats(), ora.stats(), "episode statistics")
    eng.close()
    ora.close()


@pytest.mark.parametrize("sk", [0, 1, 2])
@pytest.mark.parametrize("rk", range(11))
@pytest.mark.parametrize("shape", ["wave", "v2_multi_tr"])
def test_all_fused_plugin_pairs(sk, rk, shape):
    """Every (state, reward) combination, not only the three shipped pairings: the fast path (rewards 0..2 compiled in, 3..10
    selected at run time) and the general kernel (several transformers: the transformer-0 limit and the overload sum differ)."""
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate
    from oracle.oracle import Oracle
    E, P = 21, 20
    if shape == "v2_multi_tr" and rk < 3 and sk != 0:
        pytest.skip("covered by the general-kernel shapes test")
    batch = generate(GenConfig.v2g_profit_plus_loads(E, P, 1 if shape == "wave" else 3, seed=40 + 3 * sk + rk, power_setpoint_enabled=True,
                                                     transformer_max_power=60.0))
    eng = _engine(batch, rk, sk, flags=4)
    assert eng.kernel_name == _expected_kernel(batch, sk, rk)
    ora = Oracle(batch, rk, sk)
    D, T = eng.D, eng.T
    d_act, d_obs, d_rew = eng.empty((T, E, P)), eng.empty((T, E, D)), eng.empty((T, E))
    eng.fill_uniform(d_act, T * E * P, 3, -1.0, 1.0)
    acts = host_uniform(T * E * P, 3, -1.0, 1.0).reshape(T, E, P)
    eng.reset()
    eng.step_n(T, d_act, E * P, d_obs, E * D, d_rew, E, None, 0, None, 0, auto_reset=False, persistent=True)
    obs, rew = d_obs.to_host(), d_rew.to_host()
    ora.reset()
    for t in range(T):
        o, r, d, m, rc = ora.step(acts[t].copy())
        _close(obs[t], o, f"obs[{t}]")
        _close(rew[t], r, f"reward[{t}]")
    _close(eng.stats(), ora.stats(), "episode statistics")
    eng.close()
    ora.close()
