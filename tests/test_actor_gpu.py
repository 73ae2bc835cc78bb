"""The policy-in-the-loop path (BASELINE configs[4] shape): the fused three-layer actor kernel against a numpy forward, and
rollouts that alternate it with env steps (ev2g_rollout) against the CPU oracle fed the same float32 actions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pool(kind, E, seed):
    from ev2gym_amd import _abi
    from ev2gym_amd.scenario_gen import GenConfig, generate
    if kind == "v2gppl":
        return (generate(GenConfig.v2g_profit_plus_loads(E, 50, 1, seed=seed)), _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"],
                _abi.STATE_KINDS["V2G_profit_max_loads"], -1.0)
    return generate(GenConfig.public_pst(E, 20, seed=seed)), _abi.REWARD_KINDS["SquaredTrackingErrorReward"], _abi.STATE_KINDS["PublicPST"], 0.0


@pytest.mark.parametrize("n_rows,d_in,h1,h2,d_out,lo", [(100, 162, 400, 300, 50, -1.0), (4096, 162, 400, 300, 50, -1.0), (77, 63, 400, 300, 20, 0.0),
                                                       (33, 17, 40, 70, 3, -1.0)])
def test_fused_mlp_forward_matches_numpy(n_rows, d_in, h1, h2, d_out, lo):
    """ev2g_mlp_forward == the same network in numpy with the kernel's operand rounding (bf16 operands, fp32 accumulation) to
    accumulation-order accuracy, and == the plain float32 network to bf16 accuracy.  Random (asymmetric) weights, row counts
    that do not fill the last 32-row tile, layer widths that do not fill the last 32-column tile."""
    from ev2gym_amd.actor import init_mlp_weights, mlp_forward_numpy
    from ev2gym_amd.engine import Engine
    pool, rk, sk, _ = _pool("v2gppl", 8, 1)
    eng = Engine(pool, rk, sk, device=0)
    rng = np.random.default_rng(d_in + n_rows)
    w = init_mlp_weights(d_in, d_out, seed=3, h1=h1, h2=h2)
    x = (rng.normal(0, 1, (n_rows, d_in)) * rng.uniform(0.1, 3.0, d_in)).astype(np.float32)
    m = eng.mlp_create(*w, out_lo=lo)
    dx, dy = eng.empty((n_rows, d_in), np.float32).upload(x), eng.empty((n_rows, d_out), np.float32)
    eng.mlp_forward(m, dx, dy, n_rows)
    y = dy.to_host()
    ref_bf16 = mlp_forward_numpy(x, w, lo, bf16=True)
    ref_f32 = mlp_forward_numpy(x, w, lo, bf16=False)
    assert np.abs(y - ref_bf16).max() <= 3e-3, np.abs(y - ref_bf16).max()
    assert np.abs(y - ref_f32).max() <= 6e-2, np.abs(y - ref_f32).max()
    assert y.min() >= lo - 1e-6 and y.max() <= 1.0 + 1e-6
    eng.mlp_destroy(m)
    eng.close()


@pytest.mark.parametrize("kind", ["v2gppl", "pst"])
def test_rollout_with_the_fused_actor_matches_oracle(kind):
    """actor(obs_f32) -> actions_f32 -> step, alternating on the device: every env step equals the oracle's step on the
    float32 actions the actor produced (widened to float64 on entry, like every action), and ev2g_rollout(k) == k times
    (ev2g_mlp_forward, ev2g_step) bit for bit."""
    from ev2gym_amd.actor import FusedMLPActor
    from ev2gym_amd.engine import Engine
    from oracle.oracle import Oracle
    E = 70
    pool, rk, sk, lo = _pool(kind, E, 4)
    res = []
    for fused_call in (False, True):
        eng = Engine(pool, rk, sk, device=0, flags=4)
        P, D, T = eng.P, eng.D, eng.T
        actor = FusedMLPActor(eng, E, P, D, lo, dev=None, seed=11)
        d_rew = eng.empty((T, E))
        eng.reset()
        if fused_call:
            eng.rollout(actor.mlp, T, d_rew, E)
            res.append((actor.obs32.to_host(), d_rew.to_host(), eng.stats()))
        else:
            ora = Oracle(pool, rk, sk)
            o0 = ora.reset()
            assert np.array_equal(actor.obs32.to_host(), o0.astype(np.float32))
            for t in range(T):
                eng.mlp_forward(actor.mlp, actor.obs32, actor.act32, E)
                a32 = actor.act32.to_host()
                assert a32.min() >= lo and a32.max() <= 1.0
                eng.step_n(1, None, 0, None, 0, d_rew.at(t * E), 0, None, 0, None, 0, auto_reset=False, persistent=False)
                o, r, d, mk, rc = ora.step(a32.astype(np.float64))
                assert np.array_equal(actor.obs32.to_host(), o.astype(np.float32)), f"obs_f32[{t}]"
                got = d_rew.to_host()[t]
                assert np.abs(got - r).max() <= 1e-9 * max(1.0, np.abs(r).max()), f"reward[{t}]"
            res.append((actor.obs32.to_host(), d_rew.to_host(), eng.stats()))
            ora.close()
        actor.close()
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(np.nan_to_num(res[0][2]), np.nan_to_num(res[1][2]))


def test_rollout_graphs_follow_a_reload_and_a_new_actor():
    """ev2g_rollout replays captured HIP graphs; a scenario reload or a new actor must not replay stale ones: after either, the
    rollout equals the step-by-step execution again, and the segments are still replayed (graph launches keep counting)."""
    from ev2gym_amd import _abi
    from ev2gym_amd.actor import FusedMLPActor
    from ev2gym_amd.engine import Engine
    E = 40
    pool, rk, sk, lo = _pool("v2gppl", E, seed=1)
    eng = Engine(pool, rk, sk, device=0, flags=4)
    T, P, D = eng.T, eng.P, eng.D
    d_rew = eng.empty((T, E))

    def both_ways(actor):
        eng.reset()
        eng.rollout(actor.mlp, T, d_rew, E)
        fused = (actor.obs32.to_host().copy(), d_rew.to_host().copy())
        eng.reset()
        for t in range(T):
            eng.mlp_forward(actor.mlp, actor.obs32, actor.act32, E)
            eng.step_n(1, None, 0, None, 0, d_rew.at(t * E), 0, None, 0, None, 0, auto_reset=False, persistent=False)
        assert np.array_equal(fused[0], actor.obs32.to_host()) and np.array_equal(fused[1], d_rew.to_host())
        return fused

    a1 = FusedMLPActor(eng, E, P, D, lo, seed=1)
    r1 = both_ways(a1)
    n1 = eng.rollout_graph_launches
    assert n1 >= 1
    r1b = both_ways(a1)                       # same signature again: replayed, same result
    assert eng.rollout_graph_launches > n1 and np.array_equal(r1[1], r1b[1])
    a1.close()
    a2 = FusedMLPActor(eng, E, P, D, lo, seed=2)      # another actor (its weights may land where the old ones were)
    r2 = both_ways(a2)
    assert not np.array_equal(r1[1], r2[1])
    pool2, _, _, _ = _pool("v2gppl", E, seed=7)         # reload: same shapes, other scenarios
    eng.load(pool2)
    a2.close()
    a3 = FusedMLPActor(eng, E, P, D, lo, seed=2)
    r3 = both_ways(a3)
    assert not np.array_equal(r2[1], r3[1])
    a3.close()
    eng.close()
