"""Replay files (the reference's pickle of EvCityReplay, models/replay.py:10-174) -> ScenarioBatch.

Fixtures `replay_*.npz` hold the pickle bytes the reference wrote for a recorded episode A, the scenario and the
trajectory of an env B that the reference constructed FROM that pickle (ev2gym_env.py:102-116), and the tensors
EvCityReplay derived (oracle/capture_golden.py:run_replay_case).
"""
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN_DIR

from ev2gym_amd import _abi
from ev2gym_amd.replay import load_replay, read_replay_object, replay_tensors
from ev2gym_amd.scenario import ScenarioBatch

CASES = ["replay_v2gppl_p2_rand_s21", "replay_pst_rand_s22"]


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.mark.parametrize("name", CASES)
def test_replay_loads_to_the_scenario_the_reference_simulated(name):
    g = load_golden(name)
    want = ScenarioBatch.from_single(g)
    got = load_replay(bytes(g["replay_pkl"]), v2g_enabled=want.v2g_enabled)
    for f in ("n_envs", "n_steps", "timescale", "n_chargers", "ports_per_charger", "n_transformers", "horizon"):
        assert getattr(got, f) == getattr(want, f), f
    for k, _ in _abi.BATCH_ARRAYS:
        assert np.array_equal(got.arrays[k], want.arrays[k], equal_nan=True), k
    # the forecasts in a replay are the ones the recorded episode left behind: overwritten by actuals (transformer.py:178-180)
    if got.arrays["tr_inflexible_load"].any():
        T = got.n_steps
        assert np.array_equal(got.arrays["tr_load_forecast"][0, :, :T - 1], got.arrays["tr_inflexible_load"][0, :, :T - 1])


@pytest.mark.parametrize("name", CASES)
def test_oracle_on_replay_reproduces_reference_episode(name):
    from oracle.oracle import Oracle
    g = load_golden(name)
    rk, sk = _abi.REWARD_KINDS[str(g["case"][3])], _abi.STATE_KINDS[str(g["case"][2])]
    ora = Oracle(load_replay(bytes(g["replay_pkl"])), rk, sk)
    obs = ora.reset()
    assert np.array_equal(obs[0], g["trj_obs"][0])
    for t in range(len(g["act"])):
        o, r, d, m, rc = ora.step(g["act"][t][None].copy())
        assert rc == 0
        np.testing.assert_allclose(o[0], g["trj_obs"][t + 1], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(r[0], g["trj_reward"][t], rtol=1e-12, atol=1e-12)
        assert np.array_equal(m[0], g["trj_mask"][t])
    ora.close()


@pytest.mark.parametrize("name", CASES)
def test_replay_tensors_match_the_ones_the_reference_derives(name):
    g = load_golden(name)
    t = replay_tensors(load_replay(bytes(g["replay_pkl"])))
    for k, v in t.items():
        assert np.array_equal(v, g["rep_" + k]), k


def test_replay_object_fields_and_restricted_unpickler():
    g = load_golden(CASES[0])
    r = read_replay_object(bytes(g["replay_pkl"]))
    assert r.sim_length == 112 and r.n_cs == 12 and r.max_n_ports == 2 and len(r.EVs) == len(g["scn_ev_cs"])
    assert type(r).__name__ == "EvCityReplay" and type(r.EVs[0]).__name__ == "EV"

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    with pytest.raises(pickle.UnpicklingError):
        read_replay_object(pickle.dumps(Evil()))
    with pytest.raises(pickle.UnpicklingError):
        read_replay_object(b"cbuiltins\neval\n.")
