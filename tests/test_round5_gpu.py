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
    pool = generate_native(GenConfig.v2g_profit_plus_loads(M, C, 1, seed=seed))
    eng = Engine(pool, _abi.REWARD_KINDS[reward], _abi.STATE_KINDS[state], flags=_abi.FLAG_LOG_SOC, n_active_envs=E)
    return eng, pool


@pytest.mark.parametrize("state,E,C", [("V2G_profit_max_loads", 37, 50), ("V2G_profit_max_loads", 16, 64), ("V2G_profit_max", 21, 40)])
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
        eng, pool = _engine(E, 2 * E, C, 5, state)
        P, D, T = eng.P, eng.D, eng.T
        mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=9), out_lo=-1.0)
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


def test_fused_rollout_through_the_hand_over_buffers_equals_the_two_kernel_chain(monkeypatch):
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
        mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=10), out_lo=-1.0)
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
