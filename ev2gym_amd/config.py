"""YAML scenario configs: the reference's schema is kept verbatim (example_config_files/*.yaml;
keys read in ev2gym_env.py:65-166, loaders.py, utils.py, transformer.py)."""
from __future__ import annotations

import yaml

from .scenario_gen import GenConfig


def load_yaml(path_or_dict) -> dict:
    if isinstance(path_or_dict, dict):
        return path_or_dict
    with open(path_or_dict, "r") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def gen_config_from_yaml(cfg, n_envs: int, seed: int = 0) -> GenConfig:
    """Map the reference YAML keys onto the vectorised generator's config."""
    c = load_yaml(cfg)
    topo = c.get("charging_network_topology", "None")
    if topo not in (None, "None"):
        raise NotImplementedError("charging_network_topology files (heterogeneous chargers) are not supported yet")
    if c.get("simulate_grid", False):
        raise NotImplementedError("simulate_grid: True is outside the accelerated path (SURVEY.md §2 row 14)")
    cs, ev = c["charging_station"], c["ev"]
    specs = str(c.get("ev_specs_file", ""))
    return GenConfig(
        n_envs=n_envs, simulation_length=int(c["simulation_length"]), timescale=int(c["timescale"]),
        number_of_charging_stations=int(c["number_of_charging_stations"]),
        number_of_ports_per_cs=int(c["number_of_ports_per_cs"]),
        number_of_transformers=int(c["number_of_transformers"]), scenario=c["scenario"],
        spawn_multiplier=float(c["spawn_multiplier"]), hour=int(c["hour"]), v2g_enabled=bool(c["v2g_enabled"]),
        discharge_price_factor=float(c["discharge_price_factor"]),
        power_setpoint_enabled=bool(c["power_setpoint_enabled"]),
        power_setpoint_flexiblity=float(c["power_setpoint_flexiblity"]),
        inflexible_loads=bool(c["inflexible_loads"]["include"]), solar_power=bool(c["solar_power"]["include"]),
        demand_response=bool(c["demand_response"]["include"]),
        heterogeneous_ev_specs=bool(c["heterogeneous_ev_specs"]),
        fleet_with_efficiency_tables=("v2g_enabled2024" in specs),
        fleet=("ev_plus_phev" if "phev" in specs else "v2g2024"),
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
