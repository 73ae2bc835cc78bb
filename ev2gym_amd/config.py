"""YAML scenario configs: the reference's schema is kept verbatim (example_config_files/*.yaml;
keys read in ev2gym_env.py:65-166, loaders.py, utils.py, transformer.py)."""
from __future__ import annotations

import json
import os
import warnings

import numpy as np
import yaml

from .scenario_gen import GenConfig


def load_yaml(path_or_dict) -> dict:
    if isinstance(path_or_dict, dict):
        return path_or_dict
    with open(path_or_dict, "r") as f:
        c = yaml.load(f, Loader=yaml.FullLoader)
    c["_config_dir"] = os.path.dirname(os.path.abspath(path_or_dict))   # where relative file names of the config are looked up as well
    return c


def load_topology(path, search_dirs=()) -> dict:
    """A charging_network_topology JSON (example_config_files/charging_topology_10.json; parsed by the reference in
    loaders.py:259-276 for the transformers and :312-340 for the chargers) as the generator's per-charger arrays.  Chargers
    are numbered in file order, a charger's transformer is the index of the transformer entry it sits under."""
    cands = [path] + [os.path.join(d, os.path.basename(path)) for d in search_dirs]
    found = next((p for p in cands if os.path.isfile(p)), None)
    if found is None:
        raise FileNotFoundError(path)
    with open(found) as f:
        topo = json.load(f)
    keys = ("min_charge_current", "max_charge_current", "min_discharge_current", "max_discharge_current", "voltage", "phases", "n_ports")
    out = {k: [] for k in keys}
    out.update(transformer=[], tr_max_power=[])
    for i, tr in enumerate(topo.values()):
        out["tr_max_power"].append(float(tr["max_power"]))
        for ch in tr["charging_stations"].values():
            if str(ch.get("charger_type", "AC")) != "AC":
                raise NotImplementedError("charger_type 'DC': the reference's EV.step raises NotImplementedError for DC chargers too (ev.py:148-149)")
            for k in keys:
                out[k].append(ch[k])
            out["transformer"].append(i)
    return {k: np.asarray(v) for k, v in out.items()}


_BUILTIN_FLEETS = {"ev_specs_v2g_enabled2024.json": ("v2g2024", True), "ev_specs_ev_plus_phev.json": ("ev_plus_phev", False),
                   "ev_specs.json": ("ev_plus_phev", False)}


def load_ev_specs(path) -> dict:
    """The EV-specification file a config names (`ev_specs_file`, loaders.py:25-41): one entry per car model with its
    registrations (the sampling weight), battery and AC charge / discharge power, optionally a charging-efficiency curve by
    current (`ch_current` + `3ph_ch_efficiency`), which spawn_single_EV extends to every integer current 0..100 A by the nearest
    level with a non-zero value (utils.py:268-288).  Returned in the generator's layout (GenConfig.ev_specs)."""
    with open(path) as f:
        specs = json.load(f)
    if not isinstance(specs, dict) or not specs:
        raise ValueError(f"{path}: an EV specification file is a non-empty JSON object of car models")
    n = len(specs)
    out = dict(names=list(specs), registrations=np.zeros(n), battery_capacity=np.zeros(n), max_ac_charge_power=np.zeros(n),
               max_ac_discharge_power=np.zeros(n), efficiency=np.full((n, 101), np.nan))
    for i, (name, m) in enumerate(specs.items()):
        try:
            out["registrations"][i] = float(m["number_of_registrations"])
            out["battery_capacity"][i] = float(m["battery_capacity"])
            out["max_ac_charge_power"][i] = float(m["max_ac_charge_power"])
            out["max_ac_discharge_power"][i] = float(m["max_ac_discharge_power"])
        except KeyError as ex:
            raise ValueError(f"{path}: model '{name}' lacks {ex}") from None
        if "3ph_ch_efficiency" in m:
            levels, eff = list(m["ch_current"]), list(m["3ph_ch_efficiency"])
            if len(levels) != len(eff) or not all(0 <= x <= 100 for x in eff):
                raise ValueError(f"{path}: model '{name}': ch_current / 3ph_ch_efficiency must pair up, efficiencies in 0..100")
            # utils.py:279-286 verbatim in behaviour: the dict is filled IN PLACE while the levels are walked, so a level filled earlier is a
            # candidate for the next one (the value propagates from the left neighbour; ties go to the earlier key in insertion order), and
            # the keys stay what the file holds (a level 12.5 never answers an integer current)
            tab = dict(zip(levels, (float(v) for v in eff)))
            for a in range(101):
                if a not in tab or tab[a] == 0:
                    good = [k for k, v in tab.items() if v != 0]
                    if good:
                        tab[a] = tab[min(good, key=lambda k: abs(k - a))]
            out["efficiency"][i] = [tab.get(a, 1.0) for a in range(101)]   # (EV.get reads missing levels as 1, ev.py:288)
    if out["registrations"].sum() <= 0:
        raise ValueError(f"{path}: number_of_registrations sum to zero")
    return out


def load_data_tables(data_dir, scenario) -> dict:
    """The spawn / PV tables of an EV2Gym install (`<data_dir>/distribution-of-arrival.csv` ... as shipped in ev2gym/data, read by
    loaders.py:53-86,165-171) for one scenario, in the generator's layout (GenConfig.data_tables)."""
    import csv

    def table(name):
        with open(os.path.join(data_dir, name), newline="", encoding="utf-8-sig") as f:
            rows = list(csv.reader(f))
        head = [h.strip().lower() for h in rows[0]]
        return head, rows[1:]

    def column(name, col, n, by_minutes):
        head, rows = table(name)
        alias = {"workplace": ("workplace", "work"), "private": ("private", "home"), "public": ("public",)}[col]
        j = next((i for i, h in enumerate(head) if h in alias), None)
        out = np.zeros(n)
        if j is None:   # (the weekend arrival table has no workplace column: workplaces see no weekend arrivals, utils.py:519-521)
            return out
        for r in rows:
            hh, mm = r[0].split(":")
            v = r[j] if j < len(r) else ""
            out[(int(hh) * 60 + int(mm)) // by_minutes] = float(v) if v not in ("", "NaN", "nan") else 0.0   # (fillna(0), loaders.py:77-82)
        return out
    out = dict(arrival_week=column("distribution-of-arrival.csv", scenario, 96, 15),
               arrival_weekend=column("distribution-of-arrival-weekend.csv", scenario, 96, 15),
               stay=column("mean-session-length-per.csv", scenario, 48, 30), energy=column("mean-demand-per-arrival.csv", scenario, 48, 30), pv=None)
    pv = os.path.join(data_dir, "pv_netherlands.csv")
    if os.path.isfile(pv):
        with open(pv, newline="") as f:
            rd = csv.reader(f)
            head = next(rd)
            j = head.index("electricity")
            out["pv"] = np.array([float(r[j]) for r in rd])
    return out


def _resolve_ev_specs(c, cfg, data_dir):
    """ev_specs_file -> (GenConfig.ev_specs | None, fleet, fleet_with_efficiency_tables).  The file is READ when it can be found
    (as written, next to the YAML, or in `data_dir`); the three files EV2Gym itself ships may be absent -- their built-in
    representative fleets stand in, statistically fitted to them (tests/golden/spawn_stats.json) --; any other name that cannot
    be read is an error: a user's fleet is never replaced silently."""
    name = c.get("ev_specs_file", None)
    if not c.get("heterogeneous_ev_specs", False):
        return None, "v2g2024", False
    if name in (None, "None", ""):
        name = "ev_specs.json"     # loaders.py:31-33: the packaged default
    name = str(name)
    cands = [name]
    if c.get("_config_dir"):
        cands.append(os.path.join(c["_config_dir"], os.path.basename(name)))
    if data_dir:
        cands.append(os.path.join(data_dir, os.path.basename(name)))
    found = next((p for p in cands if os.path.isfile(p)), None)
    if found is not None:
        spec = load_ev_specs(found)
        return spec, "v2g2024", bool((~np.isnan(spec["efficiency"][:, 0])).any())
    base = os.path.basename(name)
    if base in _BUILTIN_FLEETS:
        fleet, tables = _BUILTIN_FLEETS[base]
        return None, fleet, tables
    raise FileNotFoundError(f"ev_specs_file: '{name}' not found (looked in: {', '.join(cands)}); only the three files EV2Gym ships have a "
                            "built-in stand-in fleet -- pass data_dir= or fix the path")


# YAML keys that select real-world data by calendar date in the reference (prices, loads, PV of that day; weekday / weekend
# arrival tables).  The scenario generator here is synthetic and calendar-free, so they have no counterpart.
_CALENDAR_KEYS = ("year", "month", "day")


def gen_config_from_yaml(cfg, n_envs: int, seed: int = 0, data_dir=None) -> GenConfig:
    """Map the reference YAML keys onto the vectorised generator's config.  Every key of the reference's schema is either
    passed through, irrelevant off the grid-simulation path, or reported (warning / error) -- none is dropped silently.
    `data_dir` (or the YAML key / environment variable `EV2GYM_DATA_DIR`) points at the `ev2gym/data` directory of an EV2Gym
    install: its arrival / stay / energy-demand tables and PV year are then used instead of the fitted ones, and EV-spec files are
    looked up there too."""
    c = load_yaml(cfg)
    data_dir = data_dir or c.get("data_dir") or os.environ.get("EV2GYM_DATA_DIR") or None
    if data_dir and not os.path.isdir(str(data_dir)):
        raise FileNotFoundError(f"data_dir '{data_dir}' is not a directory")
    if c.get("simulate_grid", False):
        raise NotImplementedError("simulate_grid: True is outside the accelerated path (SURVEY.md §2 row 14)")
    if c["scenario"] not in ("workplace", "public", "private"):
        raise ValueError(f"scenario: '{c['scenario']}' -- the scenario generator has arrival / stay / energy tables for 'workplace', "
                         "'public' and 'private' (the reference's three, utils.py:492-528)")
    if not c.get("random_day", True):
        warnings.warn("random_day: False asks for the data of the calendar day " + "-".join(str(c.get(k)) for k in _CALENDAR_KEYS) +
                      "; the scenario generator is synthetic and calendar-free, so every reset still draws a new day", stacklevel=2)
    topology = None
    topo = c.get("charging_network_topology", "None")
    if topo not in (None, "None"):
        dirs = [c["_config_dir"]] if c.get("_config_dir") else []
        try:
            topology = load_topology(str(topo), dirs)
        except FileNotFoundError:      # ev2gym_env.py:182-186 prints this and carries on with the YAML's uniform chargers
            warnings.warn(f"Did not find file {topo}: using the YAML's number_of_charging_stations / charging_station keys", stacklevel=2)
    cs, ev = c["charging_station"], c["ev"]
    il, pv, dr = c["inflexible_loads"], c["solar_power"], c["demand_response"]
    ev_specs, fleet, fleet_tables = _resolve_ev_specs(c, cfg, data_dir)
    data_tables = load_data_tables(str(data_dir), c["scenario"]) if data_dir else None
    return GenConfig(
        ev_specs=ev_specs, data_tables=data_tables,
        n_envs=n_envs, simulation_length=int(c["simulation_length"]), timescale=int(c["timescale"]),
        number_of_charging_stations=int(c["number_of_charging_stations"]),
        number_of_ports_per_cs=int(c["number_of_ports_per_cs"]),
        number_of_transformers=int(c["number_of_transformers"]), scenario=c["scenario"],
        simulation_days=str(c.get("simulation_days", "weekdays")),
        spawn_multiplier=float(c["spawn_multiplier"]), hour=int(c["hour"]), minute=int(c.get("minute", 0)),
        random_hour=bool(c.get("random_day", True) and c.get("random_hour", False)), v2g_enabled=bool(c["v2g_enabled"]),
        topology=topology, tr_seed=int(c.get("tr_seed", -1)),
        inflexible_loads_capacity_multiplier_mean=float(il.get("inflexible_loads_capacity_multiplier_mean", 1)),
        inflexible_loads_forecast_mean=float(il.get("forecast_mean", 30)), inflexible_loads_forecast_std=float(il.get("forecast_std", 5)),
        solar_power_capacity_multiplier_mean=float(pv.get("solar_power_capacity_multiplier_mean", 1)),
        solar_power_forecast_mean=float(pv.get("forecast_mean", 20)), solar_power_forecast_std=float(pv.get("forecast_std", 5)),
        dr_events_per_day=int(dr.get("events_per_day", 1)),
        dr_event_capacity_percentage_mean=float(dr.get("event_capacity_percentage_mean", 35)),
        dr_event_capacity_percentage_std=float(dr.get("event_capacity_percentage_std", 5)),
        dr_event_length_minutes_min=int(dr.get("event_length_minutes_min", 60)),
        dr_event_length_minutes_max=int(dr.get("event_length_minutes_max", 60)),
        dr_event_start_hour_mean=float(dr.get("event_start_hour_mean", 12)), dr_event_start_hour_std=float(dr.get("event_start_hour_std", 2)),
        dr_notification_of_event_minutes=int(dr.get("notification_of_event_minutes", 60)),
        discharge_price_factor=float(c["discharge_price_factor"]),
        power_setpoint_enabled=bool(c["power_setpoint_enabled"]),
        power_setpoint_flexiblity=float(c["power_setpoint_flexiblity"]),
        inflexible_loads=bool(c["inflexible_loads"]["include"]), solar_power=bool(c["solar_power"]["include"]),
        demand_response=bool(c["demand_response"]["include"]),
        heterogeneous_ev_specs=bool(c["heterogeneous_ev_specs"]),
        fleet_with_efficiency_tables=fleet_tables, fleet=fleet,
        transformer_max_power=float(c["transformer"]["max_power"]),
        cs_min_charge_current=float(cs["min_charge_current"]), cs_max_charge_current=float(cs["max_charge_current"]),
        cs_min_discharge_current=float(cs["min_discharge_current"]),
        cs_max_discharge_current=float(cs["max_discharge_current"]), cs_voltage=float(cs["voltage"]),
        cs_phases=int(cs["phases"]), ev_battery_capacity=float(ev["battery_capacity"]),
        ev_max_ac_charge_power=float(ev["max_ac_charge_power"]), ev_min_ac_charge_power=float(ev["min_ac_charge_power"]),
        ev_max_discharge_power=float(ev["max_discharge_power"]), ev_min_discharge_power=float(ev["min_discharge_power"]),
        ev_phases=int(ev["ev_phases"]), ev_charge_efficiency=float(ev["charge_efficiency"]),
        ev_discharge_efficiency=float(ev["discharge_efficiency"]), ev_transition_soc=float(ev["transition_soc"]),
        ev_transition_soc_multiplier=float(ev.get("transition_soc_multiplier", 1)),
        ev_min_battery_capacity=float(ev["min_battery_capacity"]), ev_min_time_of_stay=int(ev["min_time_of_stay"]),
        ev_min_emergency_battery_capacity=float(ev["min_emergency_battery_capacity"]),
        ev_desired_capacity=float(ev["desired_capacity"]), seed=seed)
