"""Pins the CPU oracle (oracle/ev2g_oracle.c) against the golden vectors recorded from the reference
EV2Gym.step() (oracle/capture_golden.py).  Integer / index quantities must match bit-exactly; float64
trajectories to <=1e-12 abs (in practice they are bit-identical: same operation order, libm exp)."""
import numpy as np

from ev2gym_amd import _abi
from ev2gym_amd.scenario import resolve_ports
from oracle.oracle import Oracle

ATOL = 1e-12

PORT_KEYS = [("cap", "trj_cap"), ("energy", "trj_energy"), ("current", "trj_current"), ("tot_e", "trj_tot_e"),
             ("req_e", "trj_req_e"), ("prev_power", "trj_prev_power")]
CS_KEYS = [("cs_power", "trj_cs_power"), ("cs_amps", "trj_cs_amps"), ("cs_profits", "trj_cs_profits"),
           ("cs_e_ch", "trj_cs_e_ch"), ("cs_e_dis", "trj_cs_e_dis"), ("tr_power", "trj_tr_power"),
           ("tr_amps", "trj_tr_amps")]


def _close(a, b, what, tol=ATOL):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    assert (np.isnan(a) == np.isnan(b)).all(), f"{what}: NaN pattern (port occupancy) differs"
    scale = np.maximum(1.0, np.abs(np.nan_to_num(b)))
    err = np.nan_to_num(np.abs(a - b) / scale)
    assert err.max(initial=0.0) <= tol, f"{what}: max rel/abs err {err.max():.3e}"


def test_oracle_matches_reference_trajectory(golden):
    z, batch, rk, sk = golden
    o = Oracle(batch, rk, sk)
    obs = o.reset()
    assert obs.shape[1] == z["trj_obs"].shape[1]
    _close(obs[0], z["trj_obs"][0], "reset obs")
    nT = len(z["act"])
    for t in range(nT):
        a = z["act"][t:t + 1].copy()
        obs, rew, done, mask, rc = o.step(a)
        assert rc == 0
        _close(obs[0], z["trj_obs"][t + 1], f"obs[{t}]")
        _close(rew[0], z["trj_reward"][t], f"reward[{t}]")
        assert done[0] == z["trj_done"][t]
        assert (mask[0] == z["trj_mask"][t]).all(), f"action_mask[{t}] (arrival/departure indexing)"
        assert (a[0] == z["trj_act_after"][t]).all(), "in-place zeroing of empty-port actions"
        pk = o.peek(0)
        for k, g in PORT_KEYS + CS_KEYS:
            _close(pk[k], z[g][t], f"{k}[{t}]")
        assert (pk["cycles"] == z["trj_cycles"][t]).all()
    pk = o.peek(0)
    _close(pk["usage"][:nT], z["trj_usage"], "current_power_usage")
    _close(pk["potential"][:nT], z["trj_potential"], "charge_power_potential")
    _close(pk["tr_overload"][:, :nT].T, z["trj_tr_overload"], "tr_overload")
    assert (pk["session_port"] == z["trj_ev_port"]).all(), "first-free port assignment"
    _close(pk["session_afap"], z["trj_ev_afap"], "max_energy_AFAP")
    _close(pk["session_cap"], z["trj_ev_final_cap"], "final capacity")
    if "trj_stats" in z:
        st = o.stats()[0]
        for i, k in enumerate(_abi.STAT_NAMES):
            g = z["trj_stats"][i]
            if np.isnan(g):
                assert np.isnan(st[i]), k
            else:
                assert abs(st[i] - g) <= 1e-12 * max(1.0, abs(g)), (k, st[i], g)
    o.close()


def test_port_resolution_is_action_independent(golden):
    """The host replay of the first-free rule gives the ports the reference assigned at run time."""
    z, batch, _, _ = golden
    rp = resolve_ports(batch)
    m = z["trj_ev_port"] >= 0
    assert (rp[m] == z["trj_ev_port"][m]).all()


def test_step_after_done_is_an_error(golden):
    z, batch, rk, sk = golden
    if len(z["act"]) != batch.n_steps:
        return
    o = Oracle(batch, rk, sk)
    o.reset()
    for t in range(batch.n_steps):
        o.step(z["act"][t:t + 1].copy())
    *_, rc = o.step(z["act"][0:1].copy())
    assert rc == _abi.ERR_DONE  # `assert not self.done` (ev2gym_env.py:343)
    o.close()
