"""Round-5 additions, through the C-ABI on the GPU (each block says which review item it closes)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(E, M, C, seed, state="V2G_profit_max_loads", reward="ProfitMax_TrPenalty_UserIncentives"):
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    pool = generate_native(GenConfig.public_pst(M, C, seed=seed) if state == "PublicPST" else GenConfig.v2g_profit_plus_loads(M, C, 1, seed=seed))
    eng = Engine(pool, _abi.REWARD_KINDS[reward], _abi.STATE_KINDS[state], flags=_abi.FLAG_LOG_SOC, n_active_envs=E)
    return eng, pool


@pytest.mark.parametrize("state,E,C", [("V2G_profit_max_loads", 37, 50), ("V2G_profit_max_loads", 16, 64), ("V2G_profit_max", 21, 40),
                                       ("V2G_profit_max_loads", 19, 25), ("V2G_profit_max_loads", 33, 7), ("V2G_profit_max", 5, 22),
                                       ("PublicPST", 37, 20), ("PublicPST", 16, 11), ("PublicPST", 50, 3),    # round 6: PublicPST in the 64 -> 400 -> 300 -> 32 packing
                                       ("PublicPST", 65, 20), ("PublicPST", 33, 19)])                     # ... two envs per wavefront (32 policy rows per workgroup), ragged last workgroup / last wavefront
def test_fused_actor_and_step_launch_equals_the_two_kernel_chain(state, E, C, monkeypatch):
    """one launch per rollout segment -- the policy (obs -> 400 -> 300 -> ports, bf16 MFMA) evaluated INSIDE the step
    kernel's launch by the workgroup that steps the 16 envs whose rows it reads (ev2g_step_wave<.., 1024, true>) -- against round 4's chain of
    two launches per step (EV2G_NO_FUSED=1: ev2g_mlp3_s16 then a single-step ev2g_step_wave): every observation / action / reward / done / mask
    row of a whole episode collected in segments of mixed length, the statistics and the reset observation of the next episode, bit for bit
    (same tiles, same MFMA chains, same bf16 roundings; the env arithmetic is the same code).  Ragged batches: the last workgroup is partly empty."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights
    monkeypatch.delenv("EV2G_NO_FUSED", raising=False)

    def run(fused):
        if not fused:
            monkeypatch.setenv("EV2G_NO_FUSED", "1")
        else:
            monkeypatch.delenv("EV2G_NO_FUSED", raising=False)
        pst = state == "PublicPST"
        eng, pool = _engine(E, 2 * E, C, 5, state, "SquaredTrackingErrorReward" if pst and E != 16 else "ProfitMax_TrPenalty_UserIncentives")
        P, D, T = eng.P, eng.D, eng.T
        mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=9), out_lo=0.0 if pst else -1.0)
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


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_fused_rollout_through_the_hand_over_buffers_equals_the_two_kernel_chain(precision, monkeypatch):
    """The same for ev2g_rollout (float32 observation / action hand-over buffers with step stride 0, reward / done / mask per step): the fused
    launch overwrites the one observation row step after step, like the two-kernel chain does."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights
    E, C = 40, 50

    def run(fused):
        if not fused:
            monkeypatch.setenv("EV2G_NO_FUSED", "1")
        else:
            monkeypatch.delenv("EV2G_NO_FUSED", raising=False)
        eng, pool = _engine(E, E, C, 6)
        P, D, T = eng.P, eng.D, eng.T
        mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=10), out_lo=-1.0, precision=precision)
        obs32, act32 = eng.empty((E, D), np.float32), eng.empty((E, P), np.float32)
        rew, done, mask = eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
        eng.set_extras(obs_f32=obs32, actions_f32=act32)
        eng.reset()
        rows, t = [], 0
        for k in [3, 1, 20, 1, 1, 50, 36]:
            eng.rollout(mlp, k, rew.at(t * E), E, done.at(t * E), E, mask.at(t * E * P), E * P)
            t += k
            rows.append((obs32.to_host().copy(), act32.to_host().copy()))
        assert t == T
        spec = eng.last_launch_specialisation
        out = dict(rows=rows, rew=rew.to_host(), done=done.to_host(), mask=mask.to_host(), stats=eng.stats().copy())
        eng.check_faults()
        eng.mlp_destroy(mlp)
        eng.close()
        return spec, out

    s_two, two = run(False)
    s_one, one = run(True)
    assert s_two != 4 and s_one == 4
    for (o1, a1), (o2, a2) in zip(one["rows"], two["rows"]):
        assert np.array_equal(o1, o2) and np.array_equal(a1, a2)
    for k in ("rew", "done", "mask", "stats"):
        assert np.array_equal(one[k], two[k], equal_nan=True), k


@pytest.mark.parametrize("no_dict", [False, True])
def test_device_refill_brings_car_models_the_loaded_pool_never_held(no_dict, monkeypatch):
    """Round 5, battery-maths dictionary (DESIGN par.2): the loader builds the (car model x charger kind) entries from the sessions it is handed;
    ev2g_pool_refill must add the entries of every model its fleet can draw BEFORE the device draws them.  A pool loaded from a nearly empty batch
    (four sessions, three of the fleet's models) and then refilled from a full-rate stream steps whole episodes exactly like a pool
    loaded from the host-generated scenarios; the same with EV2G_NO_DICT (one entry per session, written by the device generator)."""
    import dataclasses
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, host_uniform
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    M, S1 = 16, 41
    monkeypatch.setenv("EV2G_POOL_SESSION_CAP", "96")
    if no_dict:
        monkeypatch.setenv("EV2G_NO_DICT", "1")
    else:
        monkeypatch.delenv("EV2G_NO_DICT", raising=False)
    cfg = GenConfig.v2g_profit_plus_loads(M, 50, 1, seed=S1)
    host = generate_native(cfg)
    sparse = generate_native(dataclasses.replace(GenConfig.v2g_profit_plus_loads(M, 50, 1, seed=977), spawn_multiplier=0.04))
    n_models_sparse = len(np.unique(sparse.arrays["ev_B"])) if sparse.n_sessions else 0
    assert sparse.n_sessions <= 12 and n_models_sparse < len(np.unique(host.arrays["ev_B"])), (sparse.n_sessions, n_models_sparse)
    rk, sk = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"]
    flags = _abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE

    def episode(eng):
        E, P, D, T = eng.E, eng.P, eng.D, eng.T
        acts = eng.empty((T, E, P)).upload(host_uniform(T * E * P, 600, -1.0, 1.0).reshape(T, E, P))
        obs, rew, done, mask = eng.empty((T, E, D)), eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
        eng.reset()
        eng.step_n(T, acts, E * P, obs, E * D, rew, E, done, E, mask, E * P, auto_reset=False, persistent=True)
        out = [obs.to_host(), rew.to_host(), mask.to_host(), np.nan_to_num(eng.stats(), nan=-7.0)]
        eng.check_faults()
        return out

    ref = Engine(host, rk, sk, flags=flags)
    want = episode(ref)
    ref.close()
    eng = Engine(sparse, rk, sk, flags=flags)
    assert eng.kernel_name.startswith("ev2g_step_wave")
    eng.pool_refill(cfg, S1, 0, 0, M)
    got = episode(eng)
    assert eng.pool_refill_overflows == 0
    for a, b in zip(got, want):
        assert np.array_equal(a, b, equal_nan=True)
    eng.close()


FUSED_SWEEP = list(range(20))


@pytest.mark.parametrize("precision", ["bf16", "fp32"])   # (fp32: the float32 policy inside the launch, last session of round 6)
@pytest.mark.parametrize("case", FUSED_SWEEP)
def test_fused_launch_randomised_shapes_equal_the_two_kernel_chain(case, precision, monkeypatch):
    """A seeded sweep over what the fused actor + step launch is eligible for -- 3..64 ports (narrow envs: one wavefront each all the same), both head-table states, the three compiled-in
    rewards, ragged env counts, random segment lengths, both action ranges -- against the two-launch chain, bit for bit (rows, statistics)."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    rng = np.random.default_rng(9000 + case + 1000 * int(os.environ.get("EV2G_FUSED_SWEEP_OFFSET", "0")))   # (tools/r6/gpu_fused_sweep.sh: the sweep under other seed offsets)
    C = int(rng.integers(3, 65))
    E = int(rng.integers(1, 70))
    state = ["V2G_profit_max_loads", "V2G_profit_max"][int(rng.integers(0, 2))]
    reward = ["ProfitMax_TrPenalty_UserIncentives", "SquaredTrackingErrorReward", "profit_maximization"][int(rng.integers(0, 3))]
    lo = [-1.0, 0.0][int(rng.integers(0, 2))]
    pool = generate_native(GenConfig.v2g_profit_plus_loads(E + int(rng.integers(0, 9)), C, 1, seed=100 + case))
    off = int(rng.integers(0, pool.n_envs))
    T = pool.n_steps
    cuts = sorted(set(int(x) for x in rng.integers(1, T, int(rng.integers(0, 6)))))
    segs = [b - a for a, b in zip([0] + cuts, cuts + [T])]

    def run(fused):
        if fused:
            monkeypatch.delenv("EV2G_NO_FUSED", raising=False)
        else:
            monkeypatch.setenv("EV2G_NO_FUSED", "1")
        eng = Engine(pool, _abi.REWARD_KINDS[reward], _abi.STATE_KINDS[state], flags=_abi.FLAG_LOG_SOC, n_active_envs=E)
        P, D = eng.P, eng.D
        mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=case), out_lo=lo, precision=precision)
        obs, act = eng.empty((T + 1, E, D), np.float32), eng.empty((T, E, P), np.float32)
        rew, done, mask = eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
        eng.reset_f32(obs, off)
        t, specs = 0, set()
        for k in segs:
            eng.collect(mlp, k, obs.at(t * E * D), act.at(t * E * P), rew.at(t * E), done.at(t * E), mask.at(t * E * P))
            specs.add(eng.last_launch_specialisation)
            t += k
        out = dict(obs=obs.to_host(), act=act.to_host(), rew=rew.to_host(), done=done.to_host(), mask=mask.to_host(), stats=eng.stats().copy())
        eng.check_faults()
        eng.mlp_destroy(mlp)
        eng.close()
        return specs, out

    s2, two = run(False)
    s1, one = run(True)
    # (a network that fits the SMALL fragment packing -- at most 64 inputs and 32 outputs: V2G_profit_max with fewer than 22 ports -- keeps two launches)
    fused = state == "V2G_profit_max_loads" or C >= 22
    assert s1 == ({4} if fused else s2) and 4 not in s2, (s1, s2, C, E, state, reward)
    for k in two:
        assert np.array_equal(one[k], two[k], equal_nan=True), (k, C, E, state, reward, segs)


@pytest.mark.parametrize("no_dict", [False, True])
def test_fused_launch_on_a_device_refilled_pool(no_dict, monkeypatch):
    """The fused actor + step launch on scenarios drawn ON THE DEVICE (port state lines, SessDyn / dictionary entries and occupancy masks written by
    ev2g_refill_kernel), with and without the battery-maths dictionary: equal to the two-kernel chain on the same refilled pool, bit for bit."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    E, M = 23, 46
    if no_dict:
        monkeypatch.setenv("EV2G_NO_DICT", "1")
    else:
        monkeypatch.delenv("EV2G_NO_DICT", raising=False)
    cfg = GenConfig.v2g_profit_plus_loads(M, 50, 1, seed=61)
    pool = generate_native(cfg)

    def run(fused):
        if fused:
            monkeypatch.delenv("EV2G_NO_FUSED", raising=False)
        else:
            monkeypatch.setenv("EV2G_NO_FUSED", "1")
        eng = Engine(pool, _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"],
                     flags=_abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE, n_active_envs=E)
        eng.pool_refill(cfg, 61, 900, 0, M)   # scenarios 900.. of the stream: none of them was loaded
        P, D, T = eng.P, eng.D, eng.T
        mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=2), out_lo=-1.0)
        obs, act = eng.empty((T + 1, E, D), np.float32), eng.empty((T, E, P), np.float32)
        rew, done, mask = eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
        eng.reset_f32(obs, 7)
        t = 0
        for k in (50, 1, 61):
            eng.collect(mlp, k, obs.at(t * E * D), act.at(t * E * P), rew.at(t * E), done.at(t * E), mask.at(t * E * P))
            t += k
        spec = eng.last_launch_specialisation
        out = dict(obs=obs.to_host(), act=act.to_host(), rew=rew.to_host(), done=done.to_host(), mask=mask.to_host(), stats=eng.stats().copy())
        eng.check_faults()
        assert eng.pool_refill_overflows == 0
        eng.mlp_destroy(mlp)
        eng.close()
        return spec, out

    s2, two = run(False)
    s1, one = run(True)
    assert s1 == 4 and s2 != 4
    assert two["mask"].any() and np.abs(two["act"]).max() > 0.05
    for k in two:
        assert np.array_equal(one[k], two[k], equal_nan=True), k


def test_fused_launch_at_the_benchmarked_size_equals_the_two_kernel_chain(monkeypatch):
    """BASELINE configs[4]'s per-GPU shard (4096 envs x 50 chargers), a whole episode as ONE fused launch (what bench.py's `rollout` record and the
    collector time) against 112 x (actor launch, step launch): a full grid of 256 workgroups x 1024 threads, every transition row, the
    statistics of every env, bit for bit."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    E = 4096
    pool = generate_native(GenConfig.v2g_profit_plus_loads(E, 50, 1, seed=77)).sorted_by_busy_window(E)

    def run(fused):
        if fused:
            monkeypatch.delenv("EV2G_NO_FUSED", raising=False)
        else:
            monkeypatch.setenv("EV2G_NO_FUSED", "1")
        eng = Engine(pool, _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"], flags=_abi.FLAG_LOG_SOC)
        P, D, T = eng.P, eng.D, eng.T
        mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=4), out_lo=-1.0)
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
