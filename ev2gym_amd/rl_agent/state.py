"""state_function plugins (mirror of the reference's ev2gym/rl_agent/state.py, same names and signature).

`f(env, *args) -> 1-D float64 array`.  The three functions below are fused into the HIP step kernel when passed
as `state_function=`; their Python bodies evaluate the same layout on the single-env facade and serve as the
host fallback / documentation of the observation layout (transformer-major over ports).
"""
import numpy as np


def PublicPST(env, *args):
    """state.py:6-63   [step/T, setpoint[step] (0 at the end), usage[step-1], per port (full?, energy, time since arrival)]"""
    s = env.current_step
    state = [s / env.simulation_length,
             env.power_setpoints[s] if s < env.simulation_length else 0.0,
             env.current_power_usage[s - 1]]
    for tr in env.transformers:
        for cs in env.charging_stations:
            if cs.connected_transformer == tr.id:
                for ev in cs.evs_connected:
                    if ev is not None:
                        state += [1 if ev.get_soc() == 1 else 0.5, ev.total_energy_exchanged, s - ev.time_of_arrival]
                    else:
                        state += [0.0, 0.0, 0.0]
    return np.array(state, dtype=np.float64)


def _v2g_head(env):
    s = env.current_step
    prices = np.abs(np.asarray(env.charge_prices)[0, s:s + 20])
    if len(prices) < 20:
        prices = np.append(prices, np.zeros(20 - len(prices)))
    return [float(s), env.current_power_usage[s - 1]] + list(prices)


def _ports_of(env, tr):
    out = []
    for cs in env.charging_stations:
        if cs.connected_transformer == tr.id:
            for ev in cs.evs_connected:
                out += [ev.get_soc(), ev.time_of_departure - env.current_step] if ev is not None else [0.0, 0.0]
    return out


def V2G_profit_max(env, *args):
    """state.py:65-106"""
    state = _v2g_head(env)
    for tr in env.transformers:
        state += _ports_of(env, tr)
    return np.array(state, dtype=np.float64)


def V2G_profit_max_loads(env, *args):
    """state.py:108-155"""
    state = _v2g_head(env)
    for tr in env.transformers:
        loads, pv = tr.get_load_pv_forecast(step=env.current_step, horizon=20)
        state += list(np.asarray(loads) - np.asarray(pv))
        state += list(tr.get_power_limits(step=env.current_step, horizon=20))
        state += _ports_of(env, tr)
    return np.array(state, dtype=np.float64)


V2G_profit_max_loads._ev2g_kind = 0
PublicPST._ev2g_kind = 1
V2G_profit_max._ev2g_kind = 2
