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


# ---- writing (ev2gym_env.py:503-510) -------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES)
def test_written_replay_round_trips_and_has_the_reference_layout(name, tmp_path):
    """write_replay -> load_replay gives the same scenario, and the file is a pickle of the reference's own classes with
    the attribute sets of the file the reference wrote (oracle/check_replay_write.py has the live reference load it and
    reproduce the fixture trajectory bit-exactly; run in the build container, it needs /root/reference)."""
    import pickletools
    from ev2gym_amd.replay import write_replay
    g = load_golden(name)
    batch = load_replay(bytes(g["replay_pkl"]))
    path = write_replay(str(tmp_path / "replay_sim_x.pkl"), batch, scenario="public", sim_name="sim_x")
    back = load_replay(path)
    for k, _ in _abi.BATCH_ARRAYS:
        assert np.array_equal(back.arrays[k], batch.arrays[k], equal_nan=True), k
    globs = {str(arg) for op, arg, _ in pickletools.genops(open(path, "rb").read()) if op.name in ("GLOBAL", "STACK_GLOBAL")}
    assert {"ev2gym.models.replay EvCityReplay", "ev2gym.models.ev EV", "ev2gym.models.ev_charger EV_Charger",
            "ev2gym.models.transformer Transformer"} <= globs
    assert not any("ev2gym_amd" in s for s in globs)
    ref, mine = read_replay_object(bytes(g["replay_pkl"])), read_replay_object(path)
    assert mine.replay_path.split("replay_")[-1].split(".")[0] == "sim_x"       # ev2gym_env.py:106-108 derives sim_name so
    for a, b in ((ref, mine), (ref.EVs[0], mine.EVs[0]), (ref.charging_stations[0], mine.charging_stations[0]),
                 (ref.transformers[0], mine.transformers[0])):
        assert set(a.__dict__) == set(b.__dict__)
    for k in ("u", "ev_arrival", "t_dep", "energy_at_arrival", "ev_max_energy", "ev_max_ch_power", "ev_max_dis_power",
              "ev_des_energy", "tra_min_amps", "voltages", "cs_transformer", "charge_prices", "discharge_prices", "power_setpoints"):
        assert np.array_equal(np.asarray(getattr(mine, k), float), np.asarray(getattr(ref, k), float)), k
    for k in ("sim_length", "n_cs", "n_transformers", "timescale", "max_n_ports", "cs_transformers", "simulate_grid"):
        assert getattr(mine, k) == getattr(ref, k), k


def test_replay_reader_refuses_classes_outside_its_whitelist():
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    with pytest.raises(pickle.UnpicklingError):
        read_replay_object(pickle.dumps(Evil()))
    with pytest.raises(pickle.UnpicklingError):   # a numpy name that is not one of the reconstruction helpers
        read_replay_object(b"cnumpy\nload\n.")
    with pytest.raises(pickle.UnpicklingError):   # an ev2gym name that is not a replay class
        read_replay_object(b"cev2gym.utilities.utils\nprint_statistics\n.")


def test_replay_of_a_topology_scenario_keeps_per_charger_ports(tmp_path):
    """Chargers with different port counts (topology file): n_ports per EV_Charger, [max_n_ports, n_cs, T] tensors, round trip."""
    from ev2gym_amd.replay import write_replay
    g = load_golden("topo_v2gppl_het_rand_s41")
    batch = ScenarioBatch.from_single(g)
    assert batch.arrays["cs_n_ports"].tolist() == [3, 2, 2, 1, 1]
    path = write_replay(str(tmp_path / "replay_sim_t.pkl"), batch)
    rep = read_replay_object(path)
    assert [c.n_ports for c in rep.charging_stations] == [3, 2, 2, 1, 1] and rep.max_n_ports == 3
    assert rep.u.shape == (3, 5, batch.n_steps) and rep.u[1:, 3:].sum() == 0     # one-port chargers only ever use port 0
    back = load_replay(path)
    for k, _ in _abi.BATCH_ARRAYS:
        assert np.array_equal(back.arrays[k], batch.arrays[k], equal_nan=True), k
    # the EV ids are the ports the first-free rule gives them in the reference's run (trj_ev_port is cumulative)
    base = batch.port_base
    for k, ev in enumerate(rep.EVs):
        if g["trj_ev_port"][k] >= 0:
            assert base[ev.location] + ev.id == g["trj_ev_port"][k]
