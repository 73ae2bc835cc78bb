"""Vectorised synthetic scenario generator: builds the tensors `EV2Gym.step()` reads for E envs at once.

The reference builds one scenario per `reset()` with Python loops and pandas look-ups
(EV_spawner utils.py:477-557, spawn_single_EV :177-345, load_transformers loaders.py:227-296 +
transformer.py:80-256, load_electricity_prices loaders.py:392-461, generate_power_setpoints
utils.py:664-757), which costs 0.16-1.2 s per env and would dominate a GPU-resident
step engine.  This module draws the same *kind* of scenario -- same structure, same constraints,
statistically matched (not bit-identical: the reference's RNG streams and CSV data sets are not
reproduced; the hour-of-day tables and fleet classes below are FITTED to summary statistics of the
reference's own resets, tests/golden/spawn_stats.json, and tests/test_host_logic.py keeps them there:
occupancy, sessions per port, stay and arrival-SoC quantiles, required energy, arrival histogram) --
for thousands of envs with numpy array operations:

  * arrivals: a Bernoulli trial per (port, step) against a time-of-day rate, with the reference's
    "port must have been empty for 3 steps" rule and the end-of-simulation cut-off;
  * EV specs: a fleet table (battery size, AC power, 3-phase efficiency-vs-current table) or homogeneous
    config values; two-stage model parameters as in spawn_single_EV;
  * prices: hourly day-ahead-like curve, negated for charging (loaders.py:439-442);
  * transformers: inflexible load + PV + forecasts + one demand-response event (transformer.py:80-256);
  * power setpoints (PublicPST): price-weighted spread of each session's energy, median-smoothed.

`step()` parity never depends on this file: tests feed identical tensors to the engine and the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _abi
from .scenario import ScenarioBatch

# A synthetic V2G-capable fleet: (share, battery kWh, max AC kW, efficiency % at 6..16 A in 2 A steps)
_FLEET_V2G = [
    (0.22, 57.5, 11.0, (87, 87, 90, 90, 90, 90)),
    (0.18, 57.5, 11.0, (87, 87, 90, 90, 90, 90)),
    (0.13, 64.8, 11.0, (90, 90, 90, 90, 90, 90)),
    (0.11, 58.0, 11.0, (87, 87, 90, 90, 90, 90)),
    (0.10, 58.0, 11.0, (87, 90, 90, 90, 90, 90)),
    (0.09, 64.0, 11.0, (90, 90, 93, 93, 93, 93)),
    (0.09, 46.3, 7.4, (84, 87, 90, 90, 90, 90)),
    (0.08, 77.0, 11.0, (90, 90, 90, 90, 90, 90)),
]


# A mixed BEV / plug-in-hybrid fleet (the PublicPST configuration): (share, battery kWh, max AC kW).  Representative
# classes fitted to the capacity distribution the reference draws for that config (39 % below 20 kWh, mean 39.7 kWh).
_FLEET_EV_PHEV = [
    (0.26, 8.0, 3.7), (0.10, 14.5, 3.7), (0.03, 39.0, 3.6), (0.045, 46.3, 7.4), (0.035, 52.0, 22.0),
    (0.32, 57.7, 11.0), (0.12, 64.5, 11.0), (0.07, 76.0, 11.0),
]


def _lut_from_levels(levels, values):
    """Nearest-non-zero fill of a {current level -> efficiency %} map over 0..100 A (utils.py:279-288)."""
    tab = np.zeros(_abi.LUT_LEN)
    lv = np.asarray(levels)
    for i in range(_abi.LUT_LEN):
        tab[i] = values[int(np.argmin(np.abs(lv - i)))]
    return tab


@dataclass
class GenConfig:
    """Mirror of the YAML keys the reference reads (V2GProfitPlusLoads.yaml / PublicPST.yaml)."""
    n_envs: int = 1
    simulation_length: int = 112
    timescale: int = 15
    number_of_charging_stations: int = 25
    number_of_ports_per_cs: int = 1
    number_of_transformers: int = 1
    scenario: str = "workplace"          # workplace | public | private
    simulation_days: str = "weekdays"    # weekdays | weekends | both (ev2gym_env.py:147-154); workplaces see no weekend arrivals (utils.py:519-521)
    spawn_multiplier: float = 5.0
    hour: int = 5
    minute: int = 0
    random_hour: bool = False             # ev2gym_env.py:131-133: start hour drawn from 5..15 (here: once per drawn batch)
    v2g_enabled: bool = True
    discharge_price_factor: float = 1.0
    power_setpoint_enabled: bool = False
    power_setpoint_flexiblity: float = 80.0
    inflexible_loads: bool = True
    solar_power: bool = True
    demand_response: bool = True
    # transformer.py:85-93,192-256 (the YAML's inflexible_loads / solar_power / demand_response sub-keys)
    inflexible_loads_capacity_multiplier_mean: float = 1.0
    inflexible_loads_forecast_mean: float = 30.0
    inflexible_loads_forecast_std: float = 5.0
    solar_power_capacity_multiplier_mean: float = 1.0
    solar_power_forecast_mean: float = 20.0
    solar_power_forecast_std: float = 5.0
    dr_events_per_day: int = 1
    dr_event_capacity_percentage_mean: float = 35.0
    dr_event_capacity_percentage_std: float = 5.0
    dr_event_length_minutes_min: int = 60
    dr_event_length_minutes_max: int = 60
    dr_event_start_hour_mean: float = 12.0
    dr_event_start_hour_std: float = 2.0
    dr_notification_of_event_minutes: int = 60
    tr_seed: int = -1                     # != -1: transformer loads / PV / events come from their own generator (ev2gym_env.py:97-100)
    heterogeneous_ev_specs: bool = True
    fleet_with_efficiency_tables: bool = True   # ev_specs_v2g_enabled2024-like; False = scalar eta in [0.95,1]
    fleet: str = "v2g2024"                      # "v2g2024" | "ev_plus_phev" (mixed BEV / plug-in hybrids, PublicPST)
    transformer_max_power: float = 100.0
    cs_min_charge_current: float = 0.0
    cs_max_charge_current: float = 32.0
    cs_min_discharge_current: float = 0.0
    cs_max_discharge_current: float = -32.0
    cs_voltage: float = 400.0
    cs_phases: int = 3
    ev_battery_capacity: float = 50.0
    ev_max_ac_charge_power: float = 11.0
    ev_min_ac_charge_power: float = 0.0
    ev_max_discharge_power: float = -11.0
    ev_min_discharge_power: float = 0.0
    ev_phases: int = 3
    ev_charge_efficiency: float = 1.0
    ev_discharge_efficiency: float = 1.0
    ev_transition_soc: float = 1.0
    ev_transition_soc_multiplier: float = 5.0
    ev_min_battery_capacity: float = 5.0
    ev_min_time_of_stay: int = 180
    ev_min_emergency_battery_capacity: float = 25.0
    ev_desired_capacity: float = 1.0
    seed: int = 0
    # charging_network_topology file (ev2gym_env.py:176-186; loaders.py:259-276,312-340): per-charger arrays "n_ports",
    # "transformer", "min_charge_current", "max_charge_current", "min_discharge_current", "max_discharge_current", "voltage",
    # "phases" [C] and "tr_max_power" [R]; overrides the number_of_* / charging_station / transformer keys above
    topology: Optional[dict] = None
    # the file `ev_specs_file` names (loaders.py:25-41; config.load_ev_specs): "registrations", "battery_capacity",
    # "max_ac_charge_power", "max_ac_discharge_power" [n models] and "efficiency" [n, 101] (percent by charging current in A, the
    # nearest-level fill of utils.py:268-288; a NaN row = no table in the file: a random scalar efficiency per EV).  When set it
    # REPLACES the built-in representative fleets selected by `fleet` / `fleet_with_efficiency_tables`.
    ev_specs: Optional[dict] = None
    # tables of an EV2Gym data directory (config.load_data_tables): "arrival_week", "arrival_weekend" [96] arrivals per port per hour
    # in percent by quarter hour (distribution-of-arrival*.csv), "stay" [48] mean stay in hours and "energy" [48] mean required energy
    # in kWh by half hour of arrival (mean-session-length-per.csv, mean-demand-per-arrival.csv), "pv" [8760] hourly PV output of a
    # year (pv_netherlands.csv).  When set they are looked up like the reference does (utils.py:199-233,505-528) INSTEAD of the fitted
    # hour-of-day tables / the synthetic sun curve below.
    data_tables: Optional[dict] = None

    @staticmethod
    def v2g_profit_plus_loads(n_envs, n_chargers=50, n_transformers=1, seed=0, **kw):
        return GenConfig(n_envs=n_envs, number_of_charging_stations=n_chargers,
                         number_of_transformers=n_transformers, seed=seed, **kw)

    @staticmethod
    def public_pst(n_envs, n_chargers=20, seed=0, **kw):
        d = dict(scenario="public", v2g_enabled=False, power_setpoint_enabled=True, inflexible_loads=False,
                 solar_power=False, demand_response=False, fleet_with_efficiency_tables=False, fleet="ev_plus_phev",
                 cs_max_charge_current=16.0, cs_max_discharge_current=0.0, ev_min_time_of_stay=60)
        d.update(kw)
        return GenConfig(n_envs=n_envs, number_of_charging_stations=n_chargers, seed=seed, **d)


# Hour-of-day tables of the spawner (the role of the reference's distribution-of-arrival / time-of-connection /
# energy-demand data, utils.py:177-345): arrivals per port per hour in percent, mean stay in hours, mean required energy
# in kWh -- for an EV arriving in that hour.  Fitted (tools/calibrate_generator.py) so that scenarios drawn with the
# shipped YAMLs reproduce the reference's summary statistics (tests/golden/spawn_stats.json: arrivals per hour, stay by
# arrival time, required energy, sessions per port, occupancy); linear interpolation between the hours.
_HOURLY = {
    "workplace": dict(
        rate=np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.023, 1.241, 4.008, 8.163, 3.732, 1.516, 1.371, 1.463, 1.412, 0.945, 0.716,
                       0.568, 0.215, 0.091, 0.0, 0.0, 0.0, 0.0]),
        stay=np.array([8.0, 8.0, 8.0, 8.0, 8.0, 7.93, 7.93, 8.63, 8.63, 7.35, 6.75, 4.34, 4.04, 3.56, 3.36, 2.41, 2.41, 2.51, 2.51,
                       2.62, 2.62, 3.0, 3.0, 3.0]),
        energy=np.full(24, 14.35)),
    "private": dict(   # home charging: evening arrivals, overnight stays
        rate=np.array([0.166, 0.166, 0.166, 0.166, 0.166, 0.03, 0.018, 0.079, 0.113, 0.426, 0.277, 0.307, 0.387, 0.587, 0.627, 0.639, 0.495,
                       0.69, 2.806, 3.265, 2.131, 1.17, 1.69, 1.514]),
        stay=np.array([8.0, 8.0, 8.0, 8.0, 8.0, 5.32, 5.32, 4.32, 4.32, 3.16, 3.66, 2.48, 4.48, 3.69, 4.69, 10.58, 10.58, 13.9, 13.4,
                       13.07, 12.07, 11.09, 10.59, 8.91]),
        energy=np.full(24, 22.0)),
    # weekend days (distribution-of-arrival-weekend.csv, the *_weekend stay / energy distributions; utils.py:366-382,519-528)
    "private_weekend": dict(
        rate=np.array([0.214, 0.214, 0.214, 0.214, 0.214, 0.021, 0.029, 0.08, 0.081, 0.231, 0.486, 0.731, 1.396, 1.606, 1.679, 1.604, 0.916,
                       1.576, 2.605, 1.286, 1.485, 1.02, 1.231, 0.526]),
        stay=np.array([8.0, 8.0, 8.0, 8.0, 8.0, 4.35, 4.35, 4.31, 4.31, 3.41, 3.41, 3.16, 4.16, 3.72, 4.72, 9.71, 10.71, 14.4, 14.4, 13.11,
                       12.11, 11.47, 10.47, 8.3]),
        energy=np.full(24, 19.37)),
    "public_weekend": dict(
        rate=np.array([0.163, 0.163, 0.163, 0.163, 0.163, 0.035, 0.053, 0.083, 0.11, 0.541, 0.957, 1.38, 1.952, 1.951, 2.029, 2.049, 2.025,
                       1.987, 1.515, 1.608, 1.253, 0.895, 0.638, 1.424]),
        stay=np.array([8.0, 8.0, 8.0, 8.0, 8.0, 5.33, 5.33, 5.63, 5.63, 4.2, 4.2, 2.86, 2.86, 3.01, 3.01, 2.82, 3.32, 7.28, 8.28, 12.0,
                       12.0, 10.83, 9.83, 12.11]),
        energy=np.full(24, 13.9)),
    "public": dict(
        rate=np.array([0.153, 0.153, 0.153, 0.153, 0.153, 0.041, 0.05, 0.156, 0.86, 2.444, 1.525, 1.113, 1.251, 1.322, 1.261, 1.221,
                       1.221, 1.272, 1.731, 2.13, 1.883, 0.649, 0.741, 0.934]),
        stay=np.array([8.0, 8.0, 8.0, 8.0, 8.0, 5.51, 5.51, 5.31, 5.31, 4.44, 4.44, 2.86, 2.86, 2.98, 2.98, 2.92, 2.92, 8.4, 8.4,
                       11.77, 11.77, 10.91, 9.91, 10.96]),
        energy=np.full(24, 14.11)),
}


def _hourly(scenario, key, hours):
    tab = _HOURLY[scenario][key]
    h = np.asarray(hours, float) % 24.0
    return np.interp(h, np.arange(25), np.append(tab, tab[0]))


def _arrival_rate(scenario, hours):
    """Arrivals per port per hour in percent, by hour of day."""
    return _hourly(scenario, "rate", hours)


def _mean_stay_hours(scenario, hours):
    return _hourly(scenario, "stay", hours)


def _mean_energy_kwh(scenario, hours):
    return _hourly(scenario, "energy", hours)


def _pv_series(pv_hourly, dt):
    """pv_netherlands.csv's hourly year at the simulation timescale, smoothed like loaders.py:178-193 (two years long)."""
    x = np.asarray(pv_hourly, float)
    if dt > 60:
        k = dt // 60
        x = x[:len(x) // k * k].reshape(-1, k).max(axis=1)
    elif dt < 60:
        x = np.repeat(x, 60 // dt)
    w = max(60 // dt, 1)
    c = np.cumsum(np.insert(x, 0, 0.0))
    n = np.minimum(np.arange(1, len(x) + 1), w)
    x = (c[1:] - c[np.arange(1, len(x) + 1) - n]) / n                      # rolling(window=w, min_periods=1).mean()
    alpha = 2.0 / (w + 1.0)                                                 # ewm(span=w, adjust=True).mean()
    num, den, out = 0.0, 0.0, np.empty_like(x)
    for i, v in enumerate(x):
        num, den = v + (1 - alpha) * num, 1 + (1 - alpha) * den
        out[i] = num / den
    return np.concatenate([out, out])


def _pv_windows(pv_hourly, dt, T, start_minute, day_of_year):
    series = _pv_series(pv_hourly, dt)
    per_day = 1440 // dt
    i0 = np.asarray(day_of_year) * per_day + start_minute // dt
    return series[i0[:, None] + np.arange(T)[None, :]]


def generate(cfg: GenConfig) -> ScenarioBatch:
    rng = np.random.default_rng(cfg.seed)
    E, T, dt = cfg.n_envs, cfg.simulation_length, cfg.timescale
    Cn, npc, R = cfg.number_of_charging_stations, cfg.number_of_ports_per_cs, cfg.number_of_transformers
    a = {}
    if cfg.topology is None:
        # ---- chargers (load_ev_charger_profiles loaders.py:342-365; load_grid :494-498) ----
        a["cs_min_charge_current"] = np.full(Cn, cfg.cs_min_charge_current)
        a["cs_max_charge_current"] = np.full(Cn, cfg.cs_max_charge_current)
        a["cs_min_discharge_current"] = np.full(Cn, cfg.cs_min_discharge_current if cfg.v2g_enabled else 0.0)
        a["cs_max_discharge_current"] = np.full(Cn, cfg.cs_max_discharge_current if cfg.v2g_enabled else 0.0)
        a["cs_voltage"] = np.full(Cn, cfg.cs_voltage)
        a["cs_phases"] = np.full(Cn, cfg.cs_phases, np.int32)
        a["cs_transformer"] = (np.arange(Cn) % R).astype(np.int32)
        a["cs_n_ports"] = np.full(Cn, npc, np.int32)
        tr_cap = np.full(R, cfg.transformer_max_power)
    else:
        # ---- chargers and transformers from the topology file (loaders.py:259-276, 312-340): values as written, v2g_enabled not consulted ----
        tp = cfg.topology
        Cn, R = len(tp["n_ports"]), len(tp["tr_max_power"])
        for k in ("min_charge_current", "max_charge_current", "min_discharge_current", "max_discharge_current", "voltage"):
            a["cs_" + k] = np.asarray(tp[k], float)
        a["cs_phases"] = np.asarray(tp["phases"], np.int32)
        a["cs_transformer"] = np.asarray(tp["transformer"], np.int32)
        a["cs_n_ports"] = np.asarray(tp["n_ports"], np.int32)
        npc = int(a["cs_n_ports"].max())
        tr_cap = np.asarray(tp["tr_max_power"], float)
    port_cs = np.repeat(np.arange(Cn), a["cs_n_ports"])    # charger of every port, ports numbered cumulatively (ev2gym_env.py:364-385)
    P = len(port_cs)
    tr_cap = tr_cap[None, :, None]

    if cfg.scenario not in ("workplace", "public", "private"):
        raise ValueError(f"scenario '{cfg.scenario}': the spawner has tables for 'workplace', 'public' and 'private'")
    if cfg.simulation_days not in ("weekdays", "weekends", "both"):
        raise ValueError(f"simulation_days '{cfg.simulation_days}': weekdays, weekends or both")
    hour = int(rng.integers(5, 16)) if cfg.random_hour else cfg.hour
    step_hours = hour + cfg.minute / 60.0 + np.arange(T + 24) * dt / 60.0          # hour-of-day (unwrapped) of every step
    hod = step_hours % 24.0

    # ---- prices (load_electricity_prices loaders.py:392-461): hourly, EUR/MWh -> EUR/kWh ----
    hour_idx = np.floor(step_hours[:T]).astype(int)
    n_hours = hour_idx.max() + 1
    base = 75 + 40 * np.sin((np.arange(n_hours) % 24 - 7) / 24 * 2 * np.pi) + 30 * np.sin((np.arange(n_hours) % 24 - 17) / 12 * 2 * np.pi)
    hourly = np.maximum(base[None, :] * rng.uniform(0.6, 1.6, (E, 1)) + rng.normal(0, 12, (E, n_hours)), 3.0)
    price = np.round(hourly, 2)[:, hour_idx] / 1000.0
    a["charge_price"] = -price
    a["discharge_price"] = price * cfg.discharge_price_factor

    # ---- EV sessions (EV_spawner utils.py:477-557) ----
    min_stay_steps = cfg.ev_min_time_of_stay // dt
    free_from = np.zeros((E, P), np.int64)       # first spawn step t at which the port passes the 3-step-empty rule
    # weekday or weekend tables per env: the reference's date decides (workplaces are always simulated on weekdays, ev2gym_env.py:141-145;
    # 'both' = a uniformly random day: two in seven are weekend days)
    if cfg.scenario == "workplace" or cfg.simulation_days == "weekdays":
        weekend = np.zeros(E, bool)
    elif cfg.simulation_days == "weekends":
        weekend = np.ones(E, bool)
    else:
        weekend = rng.random(E) < 2.0 / 7.0

    def table(fn):   # [E, len(hod)]
        wd = fn(cfg.scenario, hod)
        if not weekend.any():
            return np.broadcast_to(wd, (E, len(hod)))
        return np.where(weekend[:, None], fn(cfg.scenario + "_weekend", hod)[None, :], wd[None, :])
    if cfg.data_tables is None:
        rate = table(_arrival_rate) * (dt / 60.0) * cfg.spawn_multiplier   # percent per step
        stay_mean = table(_mean_stay_hours)
        energy_mean = table(_mean_energy_kwh)
    else:
        # the reference's own tables, looked up its way: arrivals by quarter hour (utils.py:505-528, workplaces only 06:00-18:59),
        # stay / required energy by the half hour of arrival (utils.py:199-233)
        dtab = cfg.data_tables
        minute_of_day = np.floor(hod * 60.0 + 1e-9).astype(int) % 1440
        q, hh = minute_of_day // 15, minute_of_day // 30
        arr = np.where(weekend[:, None], np.asarray(dtab["arrival_weekend"], float)[q][None, :], np.asarray(dtab["arrival_week"], float)[q][None, :])
        if cfg.scenario == "workplace":
            arr = np.where((minute_of_day // 60 < 6) | (minute_of_day // 60 > 18), 0.0, arr)
        rate = arr * (dt / 60.0) * cfg.spawn_multiplier
        stay_mean = np.broadcast_to(np.asarray(dtab["stay"], float)[hh], (E, len(hod)))
        energy_mean = np.broadcast_to(np.asarray(dtab["energy"], float)[hh], (E, len(hod)))
    spec = cfg.ev_specs
    if cfg.heterogeneous_ev_specs:
        if spec is not None:
            share = np.asarray(spec["registrations"], float)
            fleet_B = np.asarray(spec["battery_capacity"], float)
            fleet_pac = np.asarray(spec["max_ac_charge_power"], float)
        else:
            fleet = _FLEET_V2G if (cfg.fleet_with_efficiency_tables or cfg.fleet != "ev_plus_phev") else _FLEET_EV_PHEV
            share = np.array([f[0] for f in fleet])
            fleet_B = np.array([f[1] for f in fleet])
            fleet_pac = np.array([f[2] for f in fleet])
        share = share / share.sum()
    se, sp, st_, sB, spac, scap0, stdep, smodel = [], [], [], [], [], [], [], []
    for t in range(2, T - min_stay_steps - 1):
        u = rng.random((E, P)) * 100.0
        spawn = (free_from <= t) & (u < rate[:, t, None])
        if not spawn.any():
            continue
        e_idx, p_idx = np.nonzero(spawn)
        n = len(e_idx)
        req = rng.normal(energy_mean[e_idx, t], 0.5 * energy_mean[e_idx, t], n)
        req = np.where(req < 5, rng.integers(5, 10, n), req)
        if cfg.heterogeneous_ev_specs:
            model = rng.choice(len(share), n, p=share)
            B = fleet_B[model]
            pac = fleet_pac[model]
        else:
            model = np.zeros(n, int)
            B = np.full(n, cfg.ev_battery_capacity)
            pac = np.full(n, cfg.ev_max_ac_charge_power)
        cap0 = np.where(B < req, rng.integers(1, np.maximum(B.astype(int), 2), n), B - req)
        cap0 = np.where(cap0 > cfg.ev_desired_capacity * B, rng.integers(1, np.maximum(B.astype(int), 2), n), cap0)
        cap0 = np.where((cap0 < cfg.ev_min_battery_capacity) & (B > 2 * cfg.ev_min_battery_capacity),
                        cfg.ev_min_battery_capacity, cap0)
        stay = rng.normal(stay_mean[e_idx, t], 0.2 * stay_mean[e_idx, t], n) * 60.0 / dt + 1
        stay = np.maximum(stay, min_stay_steps)
        keep = ~(stay + t + 4 >= T)          # empty_ports_at_end_of_simulation (utils.py:254-256)
        e_idx, p_idx, B, pac, cap0, stay, model = e_idx[keep], p_idx[keep], B[keep], pac[keep], cap0[keep], stay[keep], model[keep]
        tdep = (stay + t + 3).astype(np.int64)
        # occupancy_list[t+1 : t_dep] = 1 and the 3-step look-back (utils.py:534-552) => next spawn step >= t_dep + 2
        free_from[e_idx, p_idx] = tdep + 2
        se.append(e_idx); sp.append(p_idx); st_.append(np.full(len(e_idx), t + 1)); sB.append(B); spac.append(pac)
        scap0.append(cap0); stdep.append(tdep); smodel.append(model)
    if se:
        se, sp, st_, sB, spac, scap0, stdep, smodel = [np.concatenate(x) for x in (se, sp, st_, sB, spac, scap0, stdep, smodel)]
    else:
        se = sp = st_ = stdep = smodel = np.zeros(0, np.int64)
        sB = spac = scap0 = np.zeros(0)
    order = np.lexsort((sp, st_, se))          # env, then arrival step, then (charger, port) order
    se, sp, st_, sB, spac, scap0, stdep, smodel = [x[order] for x in (se, sp, st_, sB, spac, scap0, stdep, smodel)]
    S = len(se)
    a["env_session_start"] = np.concatenate([[0], np.cumsum(np.bincount(se, minlength=E))]).astype(np.int64)
    a["ev_cs"] = port_cs[sp].astype(np.int32)
    a["ev_t_arr"] = st_.astype(np.int32)
    a["ev_t_dep"] = stdep.astype(np.int32)
    a["ev_cap0"] = scap0.astype(float)
    a["ev_B"] = sB
    a["ev_desired"] = cfg.ev_desired_capacity * sB
    a["ev_minB"] = np.full(S, cfg.ev_min_battery_capacity)
    a["ev_min_emerg"] = np.where(cfg.ev_min_emergency_battery_capacity > sB, 0.7 * sB, cfg.ev_min_emergency_battery_capacity)
    a["ev_pac_max"] = spac
    a["ev_tsm"] = np.full(S, cfg.ev_transition_soc_multiplier)
    if cfg.heterogeneous_ev_specs:
        a["ev_pac_min"] = np.zeros(S)
        a["ev_pdis_max"] = -spac if cfg.v2g_enabled else np.zeros(S)
        a["ev_pdis_min"] = np.zeros(S)
        a["ev_phases"] = np.full(S, 3, np.int32)
        a["ev_ts"] = np.round(0.9 - (rng.random(S) + 0.00001) / 5, 3)
        if spec is not None:
            # the file's own models: discharge power as written (utils.py:303-304: v2g_enabled is not consulted), an efficiency table
            # where the model has one (utils.py:268-288), a random scalar in [0.95, 1] where it has not (:290-296)
            a["ev_pdis_max"] = -np.asarray(spec["max_ac_discharge_power"], float)[smodel]
            eff = np.asarray(spec["efficiency"], float).reshape(len(share), _abi.LUT_LEN)
            has = ~np.isnan(eff[:, 0])
            row = np.cumsum(has) - 1
            a["lut"] = eff[has] if has.any() else np.zeros((0, _abi.LUT_LEN))
            a["ev_lut"] = np.where(has[smodel], row[smodel], -1).astype(np.int32)
            a["ev_eta_ch"] = np.where(has[smodel], np.nan, np.round(1 - (rng.random(S) + 0.00001) / 20, 3))
            a["ev_eta_dis"] = np.where(has[smodel], np.nan, np.round(1 - (rng.random(S) + 0.00001) / 20, 3))
        elif cfg.fleet_with_efficiency_tables:
            levels = [6, 8, 10, 12, 14, 16]
            a["lut"] = np.stack([_lut_from_levels(levels, f[3]) for f in _FLEET_V2G])
            a["ev_lut"] = smodel.astype(np.int32)
            a["ev_eta_ch"] = np.full(S, np.nan)
            a["ev_eta_dis"] = np.full(S, np.nan)
        else:
            a["lut"] = np.zeros((0, _abi.LUT_LEN))
            a["ev_lut"] = np.full(S, -1, np.int32)
            a["ev_eta_ch"] = np.round(1 - (rng.random(S) + 0.00001) / 20, 3)
            a["ev_eta_dis"] = np.round(1 - (rng.random(S) + 0.00001) / 20, 3)
    else:
        a["ev_pac_min"] = np.full(S, cfg.ev_min_ac_charge_power)
        a["ev_pdis_max"] = np.full(S, cfg.ev_max_discharge_power)
        a["ev_pdis_min"] = np.full(S, cfg.ev_min_discharge_power)
        a["ev_phases"] = np.full(S, cfg.ev_phases, np.int32)
        a["ev_ts"] = np.full(S, cfg.ev_transition_soc)
        a["lut"] = np.zeros((0, _abi.LUT_LEN))
        a["ev_lut"] = np.full(S, -1, np.int32)
        a["ev_eta_ch"] = np.full(S, cfg.ev_charge_efficiency)
        a["ev_eta_dis"] = np.full(S, cfg.ev_discharge_efficiency)

    # ---- transformers (loaders.py:227-296, transformer.py:38-256) ----
    if cfg.tr_seed != -1:
        rng = np.random.default_rng(cfg.tr_seed)    # the reference's tr_rng: the same loads / PV / events every episode
    maxp = np.broadcast_to(tr_cap, (E, R, T)).copy()
    minp = -maxp.copy()
    tod = (hod[:T])[None, None, :]
    if cfg.inflexible_loads:
        shape = 0.35 + 0.25 * np.sin((tod / 24.0 - 0.3) * 2 * np.pi) ** 2 + 0.5 * np.exp(-((tod / 24.0 - 0.8) / 0.08) ** 2)
        infl = shape * rng.uniform(0.6, 1.4, (E, R, 1)) + rng.normal(0, 0.03, (E, R, T))
        infl = np.abs(infl)
        mult = rng.normal(cfg.inflexible_loads_capacity_multiplier_mean, 0.1, (E, R, 1))
        infl = infl * mult * (tr_cap / infl.max(axis=2, keepdims=True) + 0.0000001)
        infl = np.clip(infl, minp, maxp)
    else:
        infl = np.zeros((E, R, T))
    if cfg.solar_power and cfg.data_tables is not None and cfg.data_tables.get("pv") is not None:
        # the reference's PV data (load_pv_generation loaders.py:165-224): the hourly series of a year brought to the simulation's
        # timescale (repeat / max-pool, rolling mean, exponentially weighted mean), the window of a random day starting at the
        # simulation's start time
        sun = _pv_windows(np.asarray(cfg.data_tables["pv"], float), dt, T, hour * 60 + cfg.minute, rng.integers(0, 365, E))[:, None, :]
        solar = -(sun * rng.uniform(0.9, 1.1, (E, R, 1))) * rng.normal(cfg.solar_power_capacity_multiplier_mean, 0.1, (E, R, 1)) * tr_cap
    elif cfg.solar_power:
        sun = np.clip(np.sin((tod - 6.5) / 13.0 * np.pi), 0, None) ** 1.5 * rng.uniform(0.3, 1.0, (E, 1, 1))
        solar = -(sun * rng.uniform(0.9, 1.1, (E, R, 1))) * rng.normal(cfg.solar_power_capacity_multiplier_mean, 0.1, (E, R, 1)) * tr_cap
        solar = np.where(tod < 24, solar, solar)
    else:
        solar = np.zeros((E, R, T))
    steps_ahead = cfg.dr_notification_of_event_minutes // dt
    n_ev = max(int(cfg.dr_events_per_day), 1) if cfg.demand_response else 1
    dr = np.zeros((E, R, n_ev, 3))
    ndr = np.zeros((E, R), np.int32)
    if cfg.demand_response:
        tt = np.arange(T)[None, None, :]
        for k in range(int(cfg.dr_events_per_day)):      # generate_demand_response_events transformer.py:80-140, one event after the other
            length = rng.integers(cfg.dr_event_length_minutes_min, cfg.dr_event_length_minutes_max + 1, (E, R))
            start_min = np.clip(rng.normal(cfg.dr_event_start_hour_mean * 60, cfg.dr_event_start_hour_std * 60, (E, R)), 0, 23 * 60)
            es = (start_min // dt - (hour * 60 + cfg.minute) // dt).astype(int)
            ee = es + length // dt
            cap = np.clip(rng.normal(cfg.dr_event_capacity_percentage_mean, cfg.dr_event_capacity_percentage_std, (E, R)), 0, 100)
            # max_power[es:ee] is a Python slice (transformer.py:118-131): negative bounds (an event that starts before the
            # simulation does) count from the END of the array; the recorded event keeps the raw bounds
            s0 = np.where(es < 0, np.maximum(es + T, 0), np.minimum(es, T))
            s1 = np.where(ee < 0, np.maximum(ee + T, 0), np.minimum(ee, T))
            inside = (tt >= s0[..., None]) & (tt < s1[..., None])
            maxp = np.where(inside, maxp - maxp * cap[..., None] / 100, maxp)
            # if the load exceeds the reduced limit inside the event, the limit is lifted to the load's maximum
            over = (inside & (infl > maxp)).any(axis=2)
            load_max = np.where(inside, infl, -np.inf).max(axis=2)
            maxp = np.where(inside & over[..., None], load_max[..., None], maxp)
            cap = np.where(over, 100 * (1 - load_max / maxp.max(axis=2)), cap)
            dr[:, :, k, 0], dr[:, :, k, 1], dr[:, :, k, 2] = es, ee, cap
        ndr[:] = int(cfg.dr_events_per_day)
    fm, fs = cfg.inflexible_loads_forecast_mean / 100, cfg.inflexible_loads_forecast_std / 100
    lf = np.clip(rng.normal(fm * infl, np.abs(fs * infl)), minp, maxp) if cfg.inflexible_loads else np.zeros((E, R, T))
    fm, fs = cfg.solar_power_forecast_mean / 100, cfg.solar_power_forecast_std / 100
    pvf = rng.normal(fm * solar, np.abs(fs * solar)) if cfg.solar_power else np.zeros((E, R, T))
    # reset() already observed step 0 (transformer.py:178-180)
    lf[:, :, 0] = infl[:, :, 0]
    pvf[:, :, 0] = solar[:, :, 0]
    a["tr_max_power"], a["tr_min_power"] = maxp, minp
    a["tr_inflexible_load"], a["tr_solar_power"] = infl, solar
    a["tr_load_forecast"], a["tr_pv_forecast"] = lf, pvf
    a["tr_dr"], a["tr_n_dr"] = dr, ndr
    a["tr_steps_ahead"] = np.full((E, R), steps_ahead, np.int32)

    # ---- power setpoints (generate_power_setpoints utils.py:664-757, simplified & vectorised) ----
    if cfg.power_setpoint_enabled and S:
        pr = np.abs(a["charge_price"])
        pr = pr / pr.max(axis=1, keepdims=True)
        sq = np.sqrt(a["cs_phases"][a["ev_cs"]])     # limits of each session's charger
        min_cs = a["cs_min_charge_current"][a["ev_cs"]] * a["cs_voltage"][a["ev_cs"]] * sq / 1000
        max_cs = a["cs_max_charge_current"][a["ev_cs"]] * a["cs_voltage"][a["ev_cs"]] * sq / 1000
        tt = np.arange(T)[None, :]
        win = (tt >= (a["ev_t_arr"][:, None] + 1)) & (tt < a["ev_t_dep"][:, None])     # steps t+2 .. t_dep-1
        w = np.abs(rng.normal(1 - pr[se], np.maximum(pr[se].min(axis=1, keepdims=True), 1e-3))) * win
        w = w / np.maximum(w.sum(axis=1, keepdims=True), 1e-12)
        need = (a["ev_B"] - a["ev_cap0"]) * (100 + cfg.power_setpoint_flexiblity) / 100
        load = w * need[:, None] * 60 / dt
        lo = np.maximum(a["ev_pac_min"], min_cs)[:, None]
        hi = np.minimum(a["ev_pac_max"], max_cs)[:, None]
        load = np.where((load > 0) & (load < lo), 0.0, np.minimum(load, hi))
        sp_ = np.zeros((E, T))
        np.add.at(sp_, se, load)
        k = 5 * max(1, int(15 / dt))
        pad = np.pad(sp_, ((0, 0), (k // 2, k - 1 - k // 2)), mode="edge")
        sp_ = np.median(np.lib.stride_tricks.sliding_window_view(pad, k, axis=1), axis=2)
        a["power_setpoints"] = sp_
    else:
        a["power_setpoints"] = np.zeros((E, T))
    return ScenarioBatch(E, T, dt, Cn, npc, R, cfg.v2g_enabled, 20, a).finalize()


def gen_config_c(cfg: GenConfig):
    """`cfg` as the C-ABI's ev2g_gen_config (include/ev2g.h) plus the numpy arrays its pointers borrow (keep them alive while the
    struct is in use): what ev2g_generate and ev2g_pool_refill take."""
    import ctypes as C
    c = _abi.GenConfigC()
    for n in _abi.GEN_INT_FIELDS:
        if n == "scenario":
            if cfg.scenario not in _abi.GEN_SCENARIOS:
                raise ValueError(f"scenario '{cfg.scenario}': the spawner has tables for 'workplace', 'public' and 'private'")
            v = _abi.GEN_SCENARIOS[cfg.scenario]
        elif n == "simulation_days":
            if cfg.simulation_days not in _abi.GEN_DAYS:
                raise ValueError(f"simulation_days '{cfg.simulation_days}': weekdays, weekends or both")
            v = _abi.GEN_DAYS[cfg.simulation_days]
        elif n == "fleet":
            v = _abi.GEN_FLEETS.get(cfg.fleet, 0)
        elif n == "n_ev_specs":
            v = 0 if cfg.ev_specs is None else len(cfg.ev_specs["registrations"])
        else:
            v = int(getattr(cfg, n))
        setattr(c, n, v)
    c.tr_seed = int(cfg.tr_seed)
    for n in _abi.GEN_DOUBLE_FIELDS:
        setattr(c, n, float(getattr(cfg, n)))
    keep = []
    if cfg.topology is not None:
        tp = cfg.topology
        c.number_of_charging_stations, c.number_of_transformers = len(tp["n_ports"]), len(tp["tr_max_power"])
        for n in _abi.GEN_TOPO_INT + _abi.GEN_TOPO_DOUBLE:
            arr = np.ascontiguousarray(tp[n[5:]], np.int32 if n in _abi.GEN_TOPO_INT else np.float64)
            keep.append(arr)
            setattr(c, n, arr.ctypes.data_as(C.POINTER(C.c_int32 if n in _abi.GEN_TOPO_INT else C.c_double)))
    if cfg.ev_specs is not None:
        sp = cfg.ev_specs
        for n, key in zip(_abi.GEN_SPEC_DOUBLE, ("registrations", "battery_capacity", "max_ac_charge_power", "max_ac_discharge_power", "efficiency")):
            arr = np.ascontiguousarray(sp[key], np.float64)
            keep.append(arr)
            setattr(c, n, arr.ctypes.data_as(C.POINTER(C.c_double)))
    if cfg.data_tables is not None:
        for n, key in zip(_abi.GEN_TABLE_DOUBLE, ("arrival_week", "arrival_weekend", "stay", "energy", "pv")):
            if cfg.data_tables.get(key) is None:
                continue
            arr = np.ascontiguousarray(cfg.data_tables[key], np.float64)
            keep.append(arr)
            setattr(c, n, arr.ctypes.data_as(C.POINTER(C.c_double)))
            if key == "pv":
                c.n_pv = len(arr)
    return c, keep


def generate_native(cfg: GenConfig, n_threads: int = 0) -> ScenarioBatch:
    """The same model drawn by the library's own generator (`ev2g_generate`, csrc/ev2g_gen.h: C++, one thread per slice of the
    scenarios, counter-based random numbers): what a non-Python host of the C-ABI uses, 20-60x faster than `generate` on a many-core box.
    Same distributions and fitted tables, different random streams -- the two agree statistically (tests/test_host_logic.py holds both to
    the reference's spawn statistics), not draw by draw.  One semantic difference: `random_hour` draws a start hour per scenario here (as the
    reference does per reset), once per batch in `generate`.  Host code only: no GPU is touched."""
    import ctypes as C
    from .engine import EngineError, load_library
    L = load_library()
    c, keep = gen_config_c(cfg)
    res = C.c_void_p()
    rc = L.ev2g_generate(C.byref(c), int(cfg.n_envs), int(cfg.seed) & (2 ** 64 - 1), int(n_threads), C.byref(res))
    if rc:
        raise EngineError(rc, (L.ev2g_last_error(None) or b"").decode())
    try:
        b = L.ev2g_gen_batch(res).contents
        E, T, R, Cn, S, ND, NL = b.n_envs, b.n_steps, b.n_transformers, b.n_chargers, int(b.n_sessions), b.n_dr_max, b.n_lut
        shapes = {"charge_price": (E, T), "discharge_price": (E, T), "power_setpoints": (E, T), "tr_n_dr": (E, R), "tr_steps_ahead": (E, R),
                  "env_session_start": (E + 1,), "tr_dr": (E, R, ND, 3), "lut": (NL, _abi.LUT_LEN)}
        a = {}
        for name, ct in _abi.BATCH_ARRAYS:
            shp = shapes.get(name) or ((E, R, T) if name.startswith("tr_") else ((Cn,) if name.startswith("cs_") else (S,)))
            n = int(np.prod(shp))
            ptr = getattr(b, name)
            a[name] = (np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shp).copy() if n else np.zeros(shp, ct))
        return ScenarioBatch(E, T, b.timescale, Cn, b.ports_per_charger, R, bool(cfg.v2g_enabled), 20, a).finalize()
    finally:
        L.ev2g_gen_free(res)


def occupancy_fraction(batch: ScenarioBatch) -> float:
    """phi: fraction of port-steps with an EV connected after the spawn phase (the action_mask mean)."""
    a = batch.arrays
    T = batch.n_steps
    # present after step t (mask[t]) for t in [t_arr-1, t_dep-1]  ->  min(t_dep, T) - (t_arr - 1) masked steps
    n = np.clip(np.minimum(a["ev_t_dep"], T) - (a["ev_t_arr"] - 1), 0, None).sum()
    return float(n) / (batch.n_envs * batch.n_ports * T)
