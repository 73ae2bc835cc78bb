"""Round-4 additions, through the C-ABI on the GPU (each block says which review item it closes)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "ev2gym_amd", "example_config_files")


def _workload(name, E=None, seed=7):
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS
    from ev2gym_amd import _abi
    from ev2gym_amd.scenario_gen import generate_native
    wl = WORKLOADS[name]
    E = E or wl["envs"]
    return wl, E, generate_native(wl["gen"](E, seed)), _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]]


@pytest.mark.parametrize("workload", ["cfg2", "cfg3"])
def test_benchmarked_launch_shape_at_full_size_equals_the_general_instantiation(workload, monkeypatch):
    """the launch the benchmark times -- ONE persistent 112-step launch of the full + wide instantiation with
    step stride 0 at BASELINE's full size -- against the general instantiation (one launch, strided outputs on a handle loaded with
    EV2G_NO_STRIDED; held to the oracle at these sizes by tests/test_engine_gpu.py): last-step outputs, all 17 statistics of every env and the
    port state of sampled envs, bit for bit.  Round 5: the same launch with every output KEPT ([T,E,*] blocks, instantiation 3:
    bench.py's `persistent_strided`) against the general one at EVERY step."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine
    wl, E, batch, rk, sk = _workload(workload)
    monkeypatch.setenv("EV2G_NO_STRIDED", "1")
    eng = Engine(batch, rk, sk, flags=_abi.FLAG_LOG_SOC)
    monkeypatch.delenv("EV2G_NO_STRIDED")
    P, D, T = eng.P, eng.D, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 321, wl["lo"], 1.0)
    sample = [0, 1, E // 3, E // 2 + 5, E - 2, E - 1]
    # general instantiation: strided outputs select it on this handle
    g_obs, g_rew, g_done, g_mask = eng.empty((T, E, D)), eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
    eng.reset()
    eng.step_n(T, acts, E * P, g_obs, E * D, g_rew, E, g_done, E, g_mask, E * P, auto_reset=False, persistent=True)
    assert eng.last_launch_specialisation == 0
    G_rew, G_done, G_mask = g_rew.to_host().copy(), g_done.to_host().copy(), g_mask.to_host().copy()
    Gh = g_obs.to_host()
    G_obs_rows = {t: Gh[t].copy() for t in (0, 1, T // 2, T - 2, T - 1)}
    del Gh
    ref = dict(obs=G_obs_rows[T - 1], rew=G_rew[-1], done=G_done[-1], mask=G_mask[-1],
               stats=eng.stats().copy(), peek=[eng.peek(e) for e in sample])
    eng.check_faults()
    eng.close()
    # the same blocks through instantiation 3 (full + wide with running output pointers)
    eng = Engine(batch, rk, sk, flags=_abi.FLAG_LOG_SOC)
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 321, wl["lo"], 1.0)
    g_obs, g_rew, g_done, g_mask = eng.empty((T, E, D)), eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
    eng.reset()
    eng.step_n(T, acts, E * P, g_obs, E * D, g_rew, E, g_done, E, g_mask, E * P, auto_reset=False, persistent=True)
    assert eng.last_launch_specialisation == 3
    assert np.array_equal(g_rew.to_host(), G_rew) and np.array_equal(g_done.to_host(), G_done) and np.array_equal(g_mask.to_host(), G_mask)
    o3 = g_obs.to_host()
    for t, row in G_obs_rows.items():
        assert np.array_equal(o3[t], row, equal_nan=True), f"obs[{t}]"
    del o3
    assert np.array_equal(eng.stats(), ref["stats"], equal_nan=True)
    g_obs.free(); g_mask.free()
    # the benchmark's launch: everything present, stride 0, one launch per episode
    obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    eng.reset()
    eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
    assert eng.last_launch_specialisation == 2
    assert np.array_equal(obs.to_host(), ref["obs"], equal_nan=True)
    assert np.array_equal(rew.to_host(), ref["rew"]) and np.array_equal(done.to_host(), ref["done"]) and np.array_equal(mask.to_host(), ref["mask"])
    assert np.array_equal(eng.stats(), ref["stats"], equal_nan=True)
    for e, pr in zip(sample, ref["peek"]):
        pk = eng.peek(e)
        assert pk.keys() == pr.keys()
        for k in pk:
            assert np.array_equal(np.asarray(pk[k]), np.asarray(pr[k]), equal_nan=True), (e, k)
    eng.check_faults()
    eng.close()


@pytest.mark.parametrize("workload,E", [("cfg2", 333), ("cfg3", 200), ("cfg4", 5)])
def test_statistics_and_reset_in_one_launch_equal_the_two_calls(workload, E):
    """ev2g_get_stats_reset (include/ev2g.h) = ev2g_get_stats, then ev2g_reset_ex: the same statistics, the same reset observation, and
    the next episode on the new pool window steps identically -- on the fast path (one and two envs per statistics wavefront) and on the general kernel."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine
    wl, E, batch, rk, sk = _workload(workload, E=2 * E)
    E //= 2

    def episode_pair(fused):
        eng = Engine(batch, rk, sk, flags=_abi.FLAG_LOG_SOC, n_active_envs=E)
        P, D, T = eng.P, eng.D, eng.T
        acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 9, wl["lo"], 1.0)
        obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
        stats = eng.empty((E, _abi.N_STATS))
        eng.reset(obs, offset=3)
        eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
        if fused:
            eng.stats_reset(stats, obs, offset=E + 1)
        else:
            eng.stats(out=stats)
            eng.reset(obs, offset=E + 1)
        assert eng.current_step == 0 and eng.scenario_offset == E + 1
        out = dict(stats1=stats.to_host().copy(), obs0=obs.to_host().copy())
        half = T // 2
        eng.step_n(half, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
        out.update(obs=obs.to_host().copy(), rew=rew.to_host().copy(), mask=mask.to_host().copy(), stats_mid=eng.stats().copy(),
                   peek=[eng.peek(e) for e in (0, E - 1)])
        eng.check_faults()
        eng.close()
        return out

    a, b = episode_pair(False), episode_pair(True)
    for k in ("stats1", "obs0", "obs", "rew", "mask", "stats_mid"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    for pa, pb in zip(a["peek"], b["peek"]):
        for k in pa:
            assert np.array_equal(np.asarray(pa[k]), np.asarray(pb[k]), equal_nan=True), k


def test_refill_overflow_counter_survives_a_reload():
    """the device generator's overflow counter lived in the scenario allocation pool, which every
    ev2g_load_scenarios frees -- load, refill, load, refill then handed the kernel a dangling pointer.  It is its own allocation now."""
    from ev2gym_amd import _abi
    from ev2gym_amd.config import gen_config_from_yaml, load_yaml
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import generate_native
    y = load_yaml(os.path.join(CFG, "V2GProfitPlusLoads.yaml"))
    g = gen_config_from_yaml(y, 48, 5)
    batch = generate_native(g)
    eng = Engine(batch, _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"],
                 flags=_abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE, n_active_envs=16)
    for rnd in range(3):
        eng.pool_refill(g, 5, 48 + 16 * rnd, 16, 16)
        assert eng.pool_refill_overflows == 0
        eng.load(generate_native(gen_config_from_yaml(y, 48, 6 + rnd)))   # frees the scenario pool
    eng.pool_refill(g, 5, 200, 0, 16)
    assert eng.pool_refill_overflows == 0
    obs = eng.empty((eng.E, eng.D))
    eng.reset(obs)
    eng.close()


def test_single_step_launches_read_state_by_the_occupancy_masks_and_agree_with_longer_launches():
    """A single-step launch fetches state lines only for the ports its scenario's occupancy / arrival masks name (step table slots 6, 7:
    ev2g_build_occ_mask_kernel at load, the device generator for refilled slots); a longer launch reads every port's line.  Mixing
    launch lengths over an episode -- 1, 1, 7, 1, 30, 1, ... -- must give what 112 single steps give, on a loaded AND on a refilled pool."""
    from ev2gym_amd import _abi
    from ev2gym_amd.config import gen_config_from_yaml, load_yaml
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import generate_native
    y = load_yaml(os.path.join(CFG, "V2GProfitPlusLoads.yaml"))
    E = 40
    g = gen_config_from_yaml(y, 2 * E, 3)
    eng = Engine(generate_native(g), _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"],
                 flags=_abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE, n_active_envs=E)
    P, D, T = eng.P, eng.D, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 4, -1.0, 1.0)
    obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    for refilled in (False, True):
        if refilled:
            eng.pool_refill(g, 3, 500, 0, 2 * E)   # other scenarios, written by the device generator (masks included)
        runs = []
        for lengths in ([1] * T, [1, 1, 7, 1, 30, 1, 2, 1, 40, 1] + [1] * 27):
            assert sum(lengths) == T
            eng.reset(obs, offset=E // 2)
            t, trace = 0, []
            for k in lengths:
                eng.step_n(k, acts.at(t * E * P), E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
                t += k
                trace.append((t, obs.to_host().copy(), rew.to_host().copy(), mask.to_host().copy()))
            runs.append((dict((tt, (o, r, m)) for tt, o, r, m in trace), eng.stats().copy(), {} if refilled else eng.peek(1)))   # (no peek into a pool the host holds no copy of)
        (a, sa, pa), (b, sb, pb) = runs
        for tt in b:
            for u, v in zip(a[tt], b[tt]):
                assert np.array_equal(u, v, equal_nan=True), (refilled, tt)
        assert np.array_equal(sa, sb, equal_nan=True)
        for k in pa:
            assert np.array_equal(np.asarray(pa[k]), np.asarray(pb[k]), equal_nan=True), k
    eng.check_faults()
    eng.close()


@pytest.mark.parametrize("kind,with_torch", [("v2gppl", False), ("pst", False), ("v2gppl", True)])
def test_device_replay_collector_against_the_oracle(kind, with_torch):
    """the off-policy collection loop of an SB3 DDPG run (train_stable_baselines.py:62-130) with the replay buffer
    ON THE DEVICE (sb3_vec_env.DeviceReplayCollector over ev2g_collect): actor forward -> env step with the transitions written in place, no
    host copy of observations.  Every transition of two episodes is held to the CPU oracle fed the very actions the actor produced: rewards,
    dones and masks like every float64 output, observations as float32 of the oracle's, terminal statistics, `terminal_observation`, and the
    reset observation of the next episode in the next block."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights, mlp_forward_numpy
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.sb3_vec_env import DeviceReplayCollector
    from ev2gym_amd.scenario_gen import GenConfig, generate
    from oracle.oracle import Oracle
    if with_torch:
        torch = pytest.importorskip("torch")
        if not torch.cuda.is_available():
            pytest.skip("torch without a GPU")
    E, M = 24, 72
    if kind == "v2gppl":
        pool, rk, sk, lo = generate(GenConfig.v2g_profit_plus_loads(M, 50, 1, seed=3)), _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"], -1.0
    else:
        pool, rk, sk, lo = generate(GenConfig.public_pst(M, 20, seed=3)), _abi.REWARD_KINDS["SquaredTrackingErrorReward"], _abi.STATE_KINDS["PublicPST"], 0.0
    stream = None
    flags = _abi.FLAG_LOG_SOC
    if with_torch:
        import torch
        torch.cuda.set_device(0)
        stream = torch.cuda.current_stream(0).cuda_stream or None
        if stream is None:
            flags |= _abi.FLAG_NULL_STREAM
    eng = Engine(pool, rk, sk, device=0, flags=flags, n_active_envs=E, stream=stream)
    P, D, T = eng.P, eng.D, eng.T
    w = init_mlp_weights(D, P, seed=5)
    col = DeviceReplayCollector(eng, w, lo, capacity_episodes=3, use_torch=with_torch)
    host = (lambda x: x.cpu().numpy()) if with_torch else (lambda x: x.to_host())
    offsets = [0]
    for ep in range(2):
        b = col.collect_episode()
        offsets.append(col.offset)
        eng.synchronize()
        assert eng.last_launch_specialisation >= 1   # the policy hand-over instantiation: the rows are read and written in place
        obs, act, rew, done, mask = host(col.obs[b]), host(col.actions[b]), host(col.reward[b]), host(col.done[b]), host(col.mask[b])
        ora = Oracle(pool.select(np.arange(offsets[ep], offsets[ep] + E) % M), rk, sk)
        o = ora.reset()
        assert np.array_equal(obs[0], o.astype(np.float32)), "reset observation (row 0)"
        for t in range(T):
            ref_a = mlp_forward_numpy(obs[t], w, lo, bf16=True)
            assert np.abs(act[t] - ref_a).max() <= 2e-2, f"action row {t}"   # (bf16 operands; the numpy mimic and the MFMA chain add in different orders)
            o, r, d, mk, rc = ora.step(act[t].astype(np.float64))
            assert rc == 0
            assert np.array_equal(obs[t + 1], o.astype(np.float32)), f"observation row {t + 1}"
            assert np.abs(rew[t] - r).max() <= 1e-9 * max(1.0, np.abs(r).max()), f"reward[{t}]"
            assert np.array_equal(done[t].astype(bool), d.astype(bool)) and np.array_equal(mask[t], mk), f"done / mask [{t}]"
        term = col.terminal_observation(b)
        assert done[T - 1].all() and np.array_equal(host(term) if with_torch else term, obs[T])
        st, so = host(col.stats), ora.stats()
        assert np.allclose(np.nan_to_num(st), np.nan_to_num(so), rtol=1e-9, atol=1e-9)
        ora.close()
    if with_torch:
        import torch
        o, a, r, n, d = col.sample(512, np.random.default_rng(0))
        assert o.shape == (512, D) and a.shape == (512, P) and n.shape == (512, D) and r.shape == (512,) and d.shape == (512,) and o.is_cuda
    col.close()
    eng.check_faults()
    eng.close()


def test_a_launch_that_falls_off_the_full_instantiation_says_which_argument_did_it():
    """the fast path's specialised instantiation is ~20 % faster and a launch loses it silently by passing strided
    outputs, a missing output, ...  The Engine now warns once and names the argument; a launch that qualifies does not warn."""
    import warnings
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate
    E = 20
    pool = generate(GenConfig.v2g_profit_plus_loads(E, 50, 1, seed=2))
    rk, sk = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"]
    eng = Engine(pool, rk, sk, flags=_abi.FLAG_LOG_SOC)
    P, D, T = eng.P, eng.D, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, -1.0, 1.0)
    obs, rew, done, mask = eng.empty((4, E, D)), eng.empty((4, E)), eng.empty((4, E), np.uint8), eng.empty((4, E, P), np.uint8)
    eng.reset()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng.step_n(2, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)   # qualifies: no warning
    assert eng.last_launch_specialisation == 2
    with warnings.catch_warnings():   # round 5: strided outputs keep the specialisation on a wide env with the SoC log (instantiation 3)
        warnings.simplefilter("error")
        eng.step_n(2, acts.at(2 * E * P), E * P, obs, E * D, rew, E, done, E, mask, E * P, auto_reset=False, persistent=True)
    assert eng.last_launch_specialisation == 3
    eng.close()
    eng = Engine(pool, rk, sk, flags=0)   # ... without the SoC log they do not: the general instantiation, and the caller is told why
    eng.reset()
    with pytest.warns(UserWarning, match="GENERAL instantiation.*stride"):
        eng.step_n(2, acts, E * P, obs, E * D, rew, E, done, E, mask, E * P, auto_reset=False, persistent=True)
    assert eng.last_launch_specialisation == 0
    eng.close()
    eng = Engine(pool, rk, sk, flags=_abi.FLAG_LOG_SOC)
    eng.reset()
    with pytest.warns(UserWarning, match="GENERAL instantiation.*NULL"):
        eng.step_n(1, acts, E * P, obs, 0, rew, 0, None, 0, mask, 0, auto_reset=False, persistent=True)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,d_in,d_out,lo", [(4096, 162, 50, -1.0), (77, 63, 20, 0.0)])
def test_float32x3_actor_is_at_the_float32_level(n_rows, d_in, d_out, lo):
    """precision="fp32x3": three bf16 terms per float32 weight and activation, six exact products per k-step on the bf16 matrix cores --
    the forward agrees with a float64 forward of the same weights to 1e-6 (the two-term mode "fp32" is held to 1e-5 by
    test_round3_gpu.test_float32_actor_matches_a_float32_forward), and "fp32" is closer to it than "bf16" by orders of magnitude."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate
    eng = Engine(generate(GenConfig.v2g_profit_plus_loads(8, 50, 1, seed=1)), _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"],
                 _abi.STATE_KINDS["V2G_profit_max_loads"], device=0)
    rng = np.random.default_rng(d_in + n_rows)
    w = init_mlp_weights(d_in, d_out, seed=3)
    x = (rng.normal(0, 1, (n_rows, d_in)) * rng.uniform(0.1, 3.0, d_in)).astype(np.float32)
    W1, b1, W2, b2, W3, b3 = [a.astype(np.float64) for a in w]
    h = np.maximum(x.astype(np.float64) @ W1.T + b1, 0)
    h = np.maximum(h @ W2.T + b2, 0)
    exact = np.tanh(h @ W3.T + b3)
    exact = exact * 0.5 + 0.5 if lo == 0.0 else exact
    err = {}
    for prec in ("fp32x3", "fp32", "bf16"):
        m = eng.mlp_create(*w, out_lo=lo, precision=prec)
        dx, dy = eng.empty((n_rows, d_in), np.float32).upload(x), eng.empty((n_rows, d_out), np.float32)
        eng.mlp_forward(m, dx, dy, n_rows)
        err[prec] = float(np.abs(dy.to_host() - exact).max())
        eng.mlp_destroy(m)
    eng.close()
    assert err["fp32x3"] <= 1e-6, err
    assert err["fp32"] <= 1e-5, err
    assert err["fp32x3"] < err["fp32"] < 0.01 * err["bf16"], err


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,d_in,d_out,lo,prec", [(37, 161, 49, -1.0, "bf16"), (4101, 170, 64, 0.0, "bf16"), (53, 33, 17, -1.0, "bf16"), (29, 64, 31, 0.0, "fp32"),
                                                       (16, 192, 50, -1.0, "fp32x3"), (8192, 63, 20, 0.0, "bf16"), (8209, 162, 50, -1.0, "bf16"), (4097, 33, 31, -1.0, "bf16")])
def test_streaming_actor_ragged_shapes(n_rows, d_in, d_out, lo, prec):
    """The 16-row streaming actor kernel at the edges of its instantiations (k-steps of 32 inputs: 161..192 / 33..64; output tiles of 16: 49..64 / 17..32):
    odd input widths (scalar input loads instead of pairs), odd output widths (scalar stores), a last workgroup with a single row, padded
    output columns that must not be written; more rows than 16 x CUs (the 32-rows-per-workgroup instantiation: 4097+ rows) -- against the numpy
    forward with the same operand rounding."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights, mlp_forward_numpy
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate
    eng = Engine(generate(GenConfig.v2g_profit_plus_loads(8, 50, 1, seed=1)), _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"],
                 _abi.STATE_KINDS["V2G_profit_max_loads"], device=0)
    rng = np.random.default_rng(d_in * 7 + n_rows)
    w = init_mlp_weights(d_in, d_out, seed=5)
    x = (rng.normal(0, 1, (n_rows, d_in)) * rng.uniform(0.1, 3.0, d_in)).astype(np.float32)
    m = eng.mlp_create(*w, out_lo=lo, precision=prec)
    dx = eng.empty((n_rows, d_in), np.float32).upload(x)
    guard = np.full((n_rows + 2, d_out), 7.0, np.float32)   # two rows behind the batch: must stay untouched
    dy = eng.empty((n_rows + 2, d_out), np.float32).upload(guard)
    eng.mlp_forward(m, dx, dy, n_rows)
    y = dy.to_host()
    assert np.all(y[n_rows:] == 7.0)
    if prec == "bf16":
        ref = mlp_forward_numpy(x, w, lo, bf16=True)
    else:   # float64 forward
        W1, b1, W2, b2, W3, b3 = [a.astype(np.float64) for a in w]
        hh = np.maximum(np.maximum(x.astype(np.float64) @ W1.T + b1, 0) @ W2.T + b2, 0)
        ref = np.tanh(hh @ W3.T + b3)
        ref = ref * 0.5 + 0.5 if lo == 0.0 else ref
    tol = 3e-3 if prec == "bf16" else (1e-5 if prec == "fp32" else 1e-6)
    assert np.abs(y[:n_rows] - ref).max() <= tol, np.abs(y[:n_rows] - ref).max()
    eng.mlp_destroy(m)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,d_in,h1,h2,d_out,lo,prec", [(300, 122, 256, 256, 25, -1.0, "bf16"), (5000, 40, 128, 64, 9, 0.0, "bf16"), (61, 150, 400, 64, 64, -1.0, "fp32"),
                                                             (45, 64, 200, 300, 32, 0.0, "fp32x3")])
def test_streaming_actor_runs_smaller_networks_zero_padded(n_rows, d_in, h1, h2, d_out, lo, prec):
    """A network that fits an instantiation of the streaming actor kernel (inputs <= 192, hidden <= 400 / 304, outputs <= 64) runs on it with the missing
    rows and columns of its weights as zeros -- e.g. SB3's [256, 256] -- and must give the numpy forward of the network itself."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import init_mlp_weights, mlp_forward_numpy
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import GenConfig, generate
    eng = Engine(generate(GenConfig.v2g_profit_plus_loads(8, 50, 1, seed=1)), _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"],
                 _abi.STATE_KINDS["V2G_profit_max_loads"], device=0)
    rng = np.random.default_rng(d_in * 3 + n_rows)
    w = init_mlp_weights(d_in, d_out, seed=9, h1=h1, h2=h2)
    x = (rng.normal(0, 1, (n_rows, d_in)) * rng.uniform(0.1, 3.0, d_in)).astype(np.float32)
    m = eng.mlp_create(*w, out_lo=lo, precision=prec)
    dx, dy = eng.empty((n_rows, d_in), np.float32).upload(x), eng.empty((n_rows, d_out), np.float32)
    eng.mlp_forward(m, dx, dy, n_rows)
    y = dy.to_host()
    if prec == "bf16":
        ref = mlp_forward_numpy(x, w, lo, bf16=True)
    else:
        W1, b1, W2, b2, W3, b3 = [a.astype(np.float64) for a in w]
        hh = np.maximum(np.maximum(x.astype(np.float64) @ W1.T + b1, 0) @ W2.T + b2, 0)
        ref = np.tanh(hh @ W3.T + b3)
        ref = ref * 0.5 + 0.5 if lo == 0.0 else ref
    tol = 3e-3 if prec == "bf16" else (1e-5 if prec == "fp32" else 1e-6)
    assert np.abs(y - ref).max() <= tol, np.abs(y - ref).max()
    eng.mlp_destroy(m)
    eng.close()


@pytest.mark.gpu
def test_device_refill_of_multi_port_chargers_has_stated_limits():
    """ev2g_pool_refill replays the first-free port assignment of multi-port chargers inside the kernel with one byte per remembered step: it refuses
    (with an error, not a wrong pool) episodes longer than 256 steps; the single-port path has no such limit."""
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine, EngineError
    from ev2gym_amd.scenario_gen import GenConfig, generate_native
    rk, sk = _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"]
    flags = _abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE
    for npc, ok in ((2, False), (1, True)):
        g = GenConfig.v2g_profit_plus_loads(6, 8, 1, seed=4, number_of_ports_per_cs=npc, simulation_length=300, timescale=5)
        eng = Engine(generate_native(g), rk, sk, device=0, flags=flags)
        if ok:
            eng.pool_refill(g, 4, 10, 0, 6)
            eng.synchronize()
            assert eng.pool_refill_overflows == 0
        else:
            with pytest.raises(EngineError):
                eng.pool_refill(g, 4, 10, 0, 6)
        eng.close()
