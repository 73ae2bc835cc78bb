"""Round 6: the big-env kernel (`ev2g_step_big`, BASELINE configs[3]) at full size over whole episodes; performance guards live in test_bench_gpu.py."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _close(a, b, what, tol=1e-9):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert (np.isnan(a) == np.isnan(b)).all(), f"{what}: NaN pattern differs"
    err = np.nan_to_num(np.abs(a - b) / np.maximum(1.0, np.abs(np.nan_to_num(b))))
    assert err.max(initial=0.0) <= tol, f"{what}: max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.mark.parametrize("no_big", [False, True], ids=["ev2g_step_big", "ev2g_step_v2_1024_spec"])
def test_cfg4_full_size_whole_episode_against_the_oracle(no_big, monkeypatch):
    """BASELINE configs[3] at its full size (2048 envs x 1000 chargers / 50 transformers), a WHOLE 112-step episode in one persistent launch with
    outputs overwritten in place (what the benchmark runs: no 7 GB observation block): every departure, arrival and statistic of the episode.
    The CPU oracle replays a sample of envs with the same actions; compared are the last step's observation / reward / done / mask, all 17
    statistics, and per sampled env the complete histories (power usage, charge-power potential, every transformer's overload), the final port
    state (capacity, energies, cycles, attached session) and every session's capacity at departure."""
    from bench import WORKLOADS
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, host_uniform
    from ev2gym_amd.scenario_gen import generate_native
    from oracle.oracle import Oracle
    wl = WORKLOADS["cfg4"]
    E = wl["envs"]
    batch = generate_native(wl["gen"](E, 11))
    rk, sk = _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]]
    monkeypatch.delenv("EV2G_NO_BIG", raising=False)
    if no_big:
        monkeypatch.setenv("EV2G_NO_BIG", "1")
    eng = Engine(batch, rk, sk, device=0, flags=_abi.FLAG_LOG_SOC)
    monkeypatch.delenv("EV2G_NO_BIG", raising=False)
    P, D, T = eng.P, eng.D, eng.T
    rng = np.random.default_rng(5)
    sample = np.unique(np.concatenate([[0, 1, 7, 8, E // 2, E - 9, E - 2, E - 1], rng.choice(E, 12, replace=False)]))
    assert len(sample) >= 16
    acts = eng.empty((T, E, P))
    a_s = np.empty((T, len(sample), P))
    for t in range(T):   # one counter stream per step: the host twin regenerates a step's block without holding 1.8 GB
        eng.fill_uniform(acts.at(t * E * P), E * P, 9000 + t, -1.0, 1.0)
        a_s[t] = host_uniform(E * P, 9000 + t, -1.0, 1.0).reshape(E, P)[sample]
    obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    eng.reset(obs)
    eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
    assert eng.last_launch_specialisation == (1 if no_big else 5), (eng.kernel_name, eng.last_launch_specialisation)
    eng.check_faults()
    ora = Oracle(batch.select(sample), rk, sk)
    ora.reset()
    for t in range(T):
        o, r, d, m, rc = ora.step(a_s[t].copy())
        assert rc == 0
    assert np.array_equal(mask.to_host()[sample], m), "action mask after the last step"
    assert np.array_equal(done.to_host()[sample], d) and d.all()
    _close(obs.to_host()[sample], o, "last observation")
    _close(rew.to_host()[sample], r, "last reward")
    st = eng.stats()
    _close(st[sample], ora.stats(), "episode statistics (17) of the sampled envs")
    assert np.isfinite(np.nan_to_num(st)).all()
    for i, e in enumerate(sample):
        pk, po = eng.peek(int(e)), ora.peek(i)
        _close(pk["power_usage"], po["usage"], f"env {e}: current_power_usage")
        _close(pk["power_potential"], po["potential"], f"env {e}: charge_power_potential")
        _close(pk["tr_overload"], po["tr_overload"], f"env {e}: tr_overload")
        _close(pk["tr_power"], po["tr_power"], f"env {e}: transformer power")
        _close(pk["port_capacity"], po["cap"], f"env {e}: capacity")
        _close(pk["port_total_energy"], po["tot_e"], f"env {e}: total energy")
        _close(pk["port_prev_power"], po["prev_power"], f"env {e}: previous power")
        _close(pk["port_energy"], po["energy"], f"env {e}: current energy")
        _close(pk["port_current"], po["current"], f"env {e}: actual current")
        assert np.array_equal(pk["port_cycles"], po["cycles"]), f"env {e}: charging cycles"
        assert np.array_equal(pk["port_session"], po["session"]), f"env {e}: attached sessions (arrival / departure indexing)"
        assert np.array_equal(pk["session_port"], po["session_port"]), f"env {e}: first-free port assignment"
        s0, s1 = batch.arrays["env_session_start"][e], batch.arrays["env_session_start"][e + 1]
        gone = batch.arrays["ev_t_dep"][s0:s1] <= T - 1   # sessions whose departure the episode processed (the engine records the capacity AT departure)
        assert gone.any()
        _close(pk["session_final_cap"][gone], po["session_cap"][gone], f"env {e}: capacity at departure")
    ora.close()
    eng.close()


BIG_SHAPES = [(513, 1), (1024, 50), (777, 7), (640, 33)]


@pytest.mark.parametrize("C,R", BIG_SHAPES, ids=[f"c{c}_r{r}" for c, r in BIG_SHAPES])
def test_big_env_kernel_every_step_against_the_oracle(C, R):
    """`ev2g_step_big` on the edges of its range (513 and 1024 ports, one and fifty transformers, uneven transformer segments): a whole episode of
    single-step launches, every output of every step against the CPU oracle; then the same episode as ONE launch (last outputs, statistics, port state)."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    from oracle.oracle import Oracle
    E = 5
    batch = generate_native(GenConfig.v2g_profit_plus_loads(E, C, R, seed=100 + C))
    rk, sk = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"]
    eng = Engine(batch, rk, sk, device=0, flags=_abi.FLAG_LOG_SOC)
    assert eng.big_kernel_reason == "", eng.big_kernel_reason
    ora = Oracle(batch, rk, sk)
    P, D, T = eng.P, eng.D, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 77, -1.0, 1.0)
    a_h = host_uniform(T * E * P, 77, -1.0, 1.0).reshape(T, E, P)
    obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    eng.reset(obs)
    ora.reset()
    for t in range(T):
        eng.step_n(1, acts.at(t * E * P), E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False)
        assert eng.last_launch_specialisation == 5
        o, r, d, m, rc = ora.step(a_h[t].copy())
        assert rc == 0
        assert np.array_equal(mask.to_host(), m), f"mask[{t}]"
        assert np.array_equal(done.to_host(), d), f"done[{t}]"
        _close(obs.to_host(), o, f"obs[{t}]")
        _close(rew.to_host(), r, f"reward[{t}]")
    st_steps = eng.stats().copy()
    _close(st_steps, ora.stats(), "episode statistics")
    pk_steps = [eng.peek(e) for e in range(E)]
    for e in range(E):
        po = ora.peek(e)
        _close(pk_steps[e]["port_capacity"], po["cap"], f"env {e}: capacity")
        assert np.array_equal(pk_steps[e]["port_cycles"], po["cycles"])
        assert np.array_equal(pk_steps[e]["port_session"], po["session"])
    eng.check_faults()
    # one launch for the whole episode: the same results (the three launch-level energy / violation totals may group their additions differently)
    eng.reset(obs)
    eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
    assert eng.last_launch_specialisation == 5
    assert np.array_equal(mask.to_host(), m) and np.array_equal(done.to_host(), d)
    _close(obs.to_host(), o, "last observation of the one-launch episode")
    assert np.allclose(eng.stats(), st_steps, rtol=1e-12, atol=1e-12, equal_nan=True)
    for e in range(E):
        pk = eng.peek(e)
        for k in ("port_capacity", "port_total_energy", "port_prev_power", "port_energy", "port_current", "power_usage", "tr_overload", "session_final_cap"):
            assert np.array_equal(pk[k], pk_steps[e][k], equal_nan=True), (e, k)
        assert np.array_equal(pk["port_cycles"], pk_steps[e]["port_cycles"]) and np.array_equal(pk["port_session"], pk_steps[e]["port_session"])
    ora.close()
    eng.close()


def test_big_env_kernel_says_why_it_does_not_apply(monkeypatch):
    """Routing is never silent: a big env that does not qualify reports the reason and runs `ev2g_step_v2<1024, 1>` (specialisation 1)."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    rk, sk = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"]
    batch = generate_native(GenConfig.v2g_profit_plus_loads(3, 600, 60, seed=1))   # 60 transformers: 1220 pair slots > 1024
    eng = Engine(batch, rk, sk, device=0, flags=_abi.FLAG_LOG_SOC)
    assert "transformers" in eng.big_kernel_reason
    E, P, D, T = eng.E, eng.P, eng.D, eng.T
    acts = eng.empty((E, P)); eng.fill_uniform(acts, E * P, 3, -1.0, 1.0)
    obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    eng.reset(obs)
    eng.step(acts, obs, rew, done, mask)
    assert eng.last_launch_specialisation == 1
    eng.close()
    monkeypatch.setenv("EV2G_NO_BIG", "1")
    batch = generate_native(GenConfig.v2g_profit_plus_loads(3, 600, 12, seed=1))
    eng = Engine(batch, rk, sk, device=0, flags=_abi.FLAG_LOG_SOC)
    assert eng.big_kernel_reason == "EV2G_NO_BIG is set"
    eng.close()


def test_big_env_kernel_on_a_device_refilled_pool(monkeypatch):
    """`ev2g_step_big` keeps a 15-entry table of the potential terms of the LOADED sessions and 16-bit windows: a pool loaded nearly empty (few sessions,
    few car models) and then re-drawn on the device brings values the table does not hold (index 15: the port's state line is fetched instead).  It must
    step whole episodes exactly like a pool loaded from the host-generated scenarios of the same stream."""
    import dataclasses
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    M, S1, C, R = 6, 53, 600, 12
    monkeypatch.setenv("EV2G_POOL_SESSION_CAP", "1400")
    cfg = GenConfig.v2g_profit_plus_loads(M, C, R, seed=S1)
    host = generate_native(cfg)
    sparse = generate_native(dataclasses.replace(GenConfig.v2g_profit_plus_loads(M, C, R, seed=977), spawn_multiplier=0.004))
    assert 0 < sparse.n_sessions <= 60, sparse.n_sessions
    rk, sk = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"]
    flags = _abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE

    def episode(eng):
        E, P, D, T = eng.E, eng.P, eng.D, eng.T
        acts = eng.empty((T, E, P)).upload(host_uniform(T * E * P, 600, -1.0, 1.0).reshape(T, E, P))
        obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
        rows = []
        eng.reset(obs)
        for t0, k in ((0, 40), (40, 1), (41, 71)):   # three launches: the table index travels through the state lines between them
            eng.step_n(k, acts.at(t0 * E * P), E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
            assert eng.last_launch_specialisation == 5, eng.big_kernel_reason
            rows.append((obs.to_host().copy(), rew.to_host().copy(), mask.to_host().copy()))
        eng.check_faults()
        return rows, np.nan_to_num(eng.stats(), nan=-7.0)

    ref = Engine(host, rk, sk, flags=flags)
    want_rows, want_stats = episode(ref)
    ref.close()
    eng = Engine(sparse, rk, sk, flags=flags)
    eng.pool_refill(cfg, S1, 0, 0, M)
    got_rows, got_stats = episode(eng)
    assert eng.pool_refill_overflows == 0
    for (o1, r1, m1), (o2, r2, m2) in zip(got_rows, want_rows):
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(m1, m2)
    assert np.array_equal(got_stats, want_stats)
    eng.close()


def test_big_env_kernel_with_several_charger_classes():
    """Chargers of different ratings (three classes of current limits, hence of power clamps): the class table in LDS against the CPU oracle."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, host_uniform
    from ev2gym_amd.scenario import ScenarioBatch
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    from oracle.oracle import Oracle
    E, C, R = 4, 700, 9
    b = generate_native(GenConfig.v2g_profit_plus_loads(E, C, R, seed=31))
    a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.arrays.items()}
    cls = np.arange(C) % 3
    for name, scale in (("cs_max_charge_current", (1.0, 0.5, 0.75)), ("cs_max_discharge_current", (1.0, 0.5, 1.0))):
        assert a[name].shape == (C,)
        a[name] = a[name] * np.asarray(scale)[cls]
    a["cs_min_charge_current"] = np.asarray([0.0, 6.0, 0.0])[cls]   # (class 1 has a minimum current: actions below it are cut to zero, ev_charger.py:167-170)
    batch = dataclasses_replace_arrays(b, a)
    rk, sk = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"]
    eng = Engine(batch, rk, sk, device=0, flags=_abi.FLAG_LOG_SOC)
    assert eng.big_kernel_reason == "", eng.big_kernel_reason
    ora = Oracle(batch, rk, sk)
    P, D, T = eng.P, eng.D, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 5, -1.0, 1.0)
    a_h = host_uniform(T * E * P, 5, -1.0, 1.0).reshape(T, E, P)
    obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    eng.reset(obs)
    ora.reset()
    for t in range(T):
        eng.step_n(1, acts.at(t * E * P), E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False)
        assert eng.last_launch_specialisation == 5
        o, r, d, m, rc = ora.step(a_h[t].copy())
        assert rc == 0 and np.array_equal(mask.to_host(), m), f"mask[{t}]"
        _close(obs.to_host(), o, f"obs[{t}]")
        _close(rew.to_host(), r, f"reward[{t}]")
    _close(eng.stats(), ora.stats(), "episode statistics")
    eng.check_faults()
    ora.close()
    eng.close()


def dataclasses_replace_arrays(batch, arrays):
    """A ScenarioBatch with the same metadata and other arrays."""
    import copy
    nb = copy.copy(batch)
    nb.arrays = arrays
    return nb


@pytest.mark.parametrize("case", range(24))
def test_randomised_device_refill_equals_the_host_generator(case):
    """Round 6 restructured ev2g_refill_kernel (spawn trials as a bit row + converged session draws, packed setpoint pairs, masks by ballot, tables a row per
    store).  A randomised sweep over what tests/test_fuzz_gpu.py draws -- shapes, timescales (96 .. 112 steps), scenarios, fleets, setpoints, demand response,
    multi-port chargers, topology files -- holds the device draw to ev2g_generate bit for bit by behaviour: a pool loaded from OTHER scenarios and refilled
    on the device steps a whole episode (observations, rewards, masks of every step, the statistics) exactly like a pool loaded from the host-generated ones."""
    import dataclasses
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, EngineError, host_uniform
    from ev2gym_amd.scenario_gen import generate_native
    from tests.test_fuzz_gpu import _draw
    rng, cfg = _draw(500 + case)
    M = 12
    S1 = int(cfg.seed)
    mk = lambda n, seed: dataclasses.replace(cfg, n_envs=n, seed=seed)   # noqa: E731
    host = generate_native(mk(M + 5, S1))
    other = generate_native(mk(M, S1 + 7919))
    if host.n_sessions == 0 or other.n_sessions == 0:
        pytest.skip("a draw without sessions")
    pst = bool(cfg.power_setpoint_enabled) and host.n_transformers == 1 and int(np.max(host.arrays.get("cs_n_ports", np.ones(1)))) == 1
    kinds = ("SquaredTrackingErrorReward", "PublicPST") if pst else ("ProfitMax_TrPenalty_UserIncentives", "V2G_profit_max_loads")
    rk, sk = _abi.REWARD_KINDS[kinds[0]], _abi.STATE_KINDS[kinds[1]]
    flags = _abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE
    lo = 0.0 if pst else -1.0

    def episode(eng):
        E, P, D, T = eng.E, eng.P, eng.D, eng.T
        act, obs, rew = eng.empty((E, P)), eng.empty((E, D)), eng.empty((E,))
        done, mask = eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
        eng.reset(obs)
        out = [obs.to_host().copy()]
        for t in range(T):
            act.upload(host_uniform(E * P, 900 + t, lo, 1.0).reshape(E, P))
            eng.step(act, obs, rew, done, mask)
            out += [obs.to_host().copy(), rew.to_host().copy(), mask.to_host().copy()]
        out.append(np.nan_to_num(eng.stats(), nan=-7.0))
        try:
            eng.check_faults()
        except EngineError:   # (multi-port chargers under out-of-range actions: the reference's over-current exception is a per-env flag here)
            pass
        return out

    eng = Engine(other, rk, sk, device=0, flags=flags)
    try:
        eng.pool_refill(mk(M, S1), S1, 3, 0, M)   # slots 0..M-1 <- scenarios 3..M+2 of the stream
    except EngineError as e:
        eng.close()
        pytest.skip(f"outside the device generator's stated limits: {e}")
    if eng.pool_refill_overflows:
        eng.close()
        pytest.skip("the refilled scenarios draw more sessions than the loaded pool's blocks hold (counted, not silent)")
    got = episode(eng)
    eng.close()
    ref_eng = Engine(host.select(np.arange(3, M + 3)), rk, sk, device=0, flags=flags)
    ref = episode(ref_eng)
    ref_eng.close()
    for i, (x, y) in enumerate(zip(got, ref)):
        assert np.array_equal(x, y), f"output {i} differs between the device-generated and the host-generated pool"


def test_page_locked_host_buffers_round_trip():
    """ev2g_host_malloc / ev2g_host_free (include/ev2g.h; Engine.pinned): page-locked host arrays as the source and the destination of the per-step copies
    (the SB3 VecEnv hand-over, ev2g_peek's staging): a round trip through device memory returns the bytes, partial copies honour the byte count, freeing
    twice or freeing foreign pointers is harmless, and what is not freed goes with the handle."""
    import ctypes as C
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario import ScenarioBatch
    z = np.load(os.path.join(ROOT, "tests", "golden", "v2gppl_c50_rand_s9.npz"))
    eng = Engine(ScenarioBatch.from_single(z).tile(4), _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"], device=0)
    a = eng.pinned((1000, 7), np.float64)
    b = eng.pinned((1000, 7), np.float64)
    assert a.shape == (1000, 7) and a.flags.c_contiguous and a.ctypes.data % 4096 == 0
    a[:] = np.random.default_rng(3).normal(size=a.shape)
    b[:] = -1.0
    d = eng.empty(a.shape)
    d.upload(a)
    eng.memcpy_d2h(b, d, 500 * 7 * 8)            # the first 500 rows only
    assert np.array_equal(b[:500], a[:500]) and (b[500:] == -1.0).all()
    eng.memcpy_d2h(b, d, a.nbytes)
    assert np.array_equal(a, b)
    eng._lib.ev2g_host_free(eng._h, C.c_void_p(a.ctypes.data))
    eng._lib.ev2g_host_free(eng._h, C.c_void_p(a.ctypes.data))      # (already freed: ignored)
    eng._lib.ev2g_host_free(eng._h, C.c_void_p(12345))               # (not ours: ignored)
    assert np.array_equal(b[3], b[3])                                # b is still valid
    eng.close()                                                      # b goes with the handle


@pytest.mark.parametrize("state,E,C,reward", [("V2G_profit_max_loads", 37, 50, "ProfitMax_TrPenalty_UserIncentives"), ("V2G_profit_max_loads", 16, 64, "SquaredTrackingErrorReward"),
                                              ("V2G_profit_max", 21, 40, "profit_maximization"), ("V2G_profit_max_loads", 19, 25, "ProfitMax_TrPenalty_UserIncentives"),
                                              ("V2G_profit_max_loads", 33, 7, "profit_maximization"), ("V2G_profit_max", 5, 22, "SquaredTrackingErrorReward"),
                                              ("PublicPST", 37, 20, "SquaredTrackingErrorReward"), ("PublicPST", 16, 11, "ProfitMax_TrPenalty_UserIncentives"), ("PublicPST", 50, 3, "profit_maximization")])
def test_fused_launch_with_the_float32_policy_equals_the_two_kernel_chain(state, E, C, reward, monkeypatch):
    """VERDICT round 5, item 4 (float32 half): the FLOAT32 policy (EV2G_MLP_F32: two bf16 terms per weight, three per activation, five MFMA products per
    k-step -- what an SB3 float32 actor computes to 1e-5) evaluated INSIDE the step kernel's launch (ev2g_step_wave<.., ACT, 1, 2> + ev2g_mlp3_inline_f32)
    against the chain of two launches per step (EV2G_NO_FUSED=1: ev2g_mlp3_s16<.., NW = 2> then a single-step launch): every observation / action /
    reward / done / mask row of a whole episode collected in segments of mixed length, the statistics and the next episode's reset observation, bit for bit
    (same tiles, same term split, same MFMA chain per tile).  Ragged batches, narrow and full-width envs, the three fused states, the three rewards."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate_native

    def run(fused):
        if fused:
            monkeypatch.delenv("EV2G_NO_FUSED", raising=False)
        else:
            monkeypatch.setenv("EV2G_NO_FUSED", "1")
        pst = state == "PublicPST"   # (one env per wavefront with this policy: 16 rows per workgroup)
        pool = generate_native(GenConfig.public_pst(2 * E, C, seed=5) if pst else GenConfig.v2g_profit_plus_loads(2 * E, C, 1, seed=5))
        eng = Engine(pool, _abi.REWARD_KINDS[reward], _abi.STATE_KINDS[state], flags=_abi.FLAG_LOG_SOC, n_active_envs=E)
        P, D, T = eng.P, eng.D, eng.T
        mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=9), out_lo=0.0 if pst else -1.0, precision="fp32")
        obs, act = eng.empty((T + 1, E, D), np.float32), eng.empty((T, E, P), np.float32)
        rew, done, mask = eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
        nxt = eng.empty((E, D), np.float32)
        stats = eng.empty((E, _abi.N_STATS))
        eng.reset_f32(obs, 3)
        t, specs = 0, set()
        for k in [1, 1, 5, 17, 1, 40, 2, 1, 30] + [1] * 14:
            assert t + k <= T
            eng.collect(mlp, k, obs.at(t * E * D), act.at(t * E * P), rew.at(t * E), done.at(t * E), mask.at(t * E * P))
            specs.add(eng.last_launch_specialisation)
            t += k
        assert t == T
        eng.stats_reset_f32(stats, nxt, 3 + E)
        eng.check_faults()
        out = dict(obs=obs.to_host(), act=act.to_host(), rew=rew.to_host(), done=done.to_host(), mask=mask.to_host(), stats=stats.to_host(), nxt=nxt.to_host())
        eng.mlp_destroy(mlp)
        eng.close()
        return specs, out

    s_two, two = run(False)
    assert 4 not in s_two
    s_one, one = run(True)
    assert s_one == {4}, s_one   # every segment ran the fused instantiation
    assert np.abs(two["act"]).max() > 0.05 and np.isfinite(two["obs"]).all()
    for k in two:
        assert np.array_equal(one[k], two[k], equal_nan=True), k
    # and the switch: EV2G_NO_FUSED_F32=1 keeps the float32 policy on two launches per step while the bf16 one stays fused
    monkeypatch.setenv("EV2G_NO_FUSED_F32", "1")
    s_off, off = run(True)
    monkeypatch.delenv("EV2G_NO_FUSED_F32")
    assert 4 not in s_off and np.array_equal(off["act"], two["act"])


@pytest.mark.parametrize("shape", ["cfg2", "cfg3"])
def test_fused_float32_policy_at_the_benchmarked_size(shape, monkeypatch):
    """BASELINE configs[4]'s per-GPU shard (4096 envs x 50 chargers; and configs[2]'s 8192 x 20 PublicPST envs, one env per wavefront: 512 workgroups) with the float32 policy: a whole episode as ONE fused launch against 112 x (actor
    launch, step launch), every action / reward / done / mask row and the statistics bit for bit; and the actions of sampled rows against a float64
    numpy forward of the same network on the float32 observations the launch wrote (the 1e-5 level the two-term weights give)."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    pst = shape == "cfg3"
    E = 8192 if pst else 4096
    pool = generate_native(GenConfig.public_pst(E, 20, seed=78) if pst else GenConfig.v2g_profit_plus_loads(E, 50, 1, seed=78)).sorted_by_busy_window(E)
    weights = init_mlp_weights(63 if pst else 162, 20 if pst else 50, seed=4)

    def run(fused):
        if fused:
            monkeypatch.delenv("EV2G_NO_FUSED", raising=False)
        else:
            monkeypatch.setenv("EV2G_NO_FUSED", "1")
        eng = Engine(pool, _abi.REWARD_KINDS["SquaredTrackingErrorReward" if pst else "ProfitMax_TrPenalty_UserIncentives"],
                     _abi.STATE_KINDS["PublicPST" if pst else "V2G_profit_max_loads"], flags=_abi.FLAG_LOG_SOC)
        P, D, T = eng.P, eng.D, eng.T
        mlp = eng.mlp_create(*weights, out_lo=0.0 if pst else -1.0, precision="fp32")
        obs, act = eng.empty((T + 1, E, D), np.float32), eng.empty((T, E, P), np.float32)
        rew, done, mask = eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
        eng.reset_f32(obs, 0)
        eng.collect(mlp, T, obs, act, rew, done, mask)
        spec = eng.last_launch_specialisation
        out = dict(act=act.to_host(), rew=rew.to_host(), done=done.to_host(), mask=mask.to_host(), stats=eng.stats().copy())
        o = obs.to_host()
        out["obs_rows"] = np.stack([o[t] for t in (0, 1, 2, T // 2, T - 1, T)])
        out["obs_sum"] = o.astype(np.float64).sum(axis=(1, 2))
        del o
        eng.check_faults()
        eng.mlp_destroy(mlp)
        eng.close()
        return spec, out

    s2, two = run(False)
    s1, one = run(True)
    assert s1 == 4 and s2 != 4
    for k in two:
        assert np.array_equal(one[k], two[k], equal_nan=True), k
    W1, b1, W2, b2, W3, b3 = [np.asarray(a, np.float64) for a in weights]
    T = one["act"].shape[0]
    for i, t in enumerate((0, 1, 2, T // 2, T - 1)):
        x = one["obs_rows"][i][::97].astype(np.float64)
        y = np.tanh(np.maximum(np.maximum(x @ W1.T + b1, 0.0) @ W2.T + b2, 0.0) @ W3.T + b3)
        if pst:
            y = 0.5 * y + 0.5   # (out_lo = 0: the action box of configs without V2G)
        assert np.abs(one["act"][t][::97] - y).max() < 5e-5, (t, np.abs(one["act"][t][::97] - y).max())
