"""Replay files: the reference's on-disk scenario format, read into a `ScenarioBatch`.

The reference saves an episode as a pickle of an `EvCityReplay` object (`ev2gym/models/replay.py:10-174`): it
holds the live `transformers`, `charging_stations` and EV profile objects plus `charge_prices`,
`discharge_prices`, `power_setpoints`, and `EV2Gym(load_from_replay_path=...)` re-runs exactly that scenario
(`ev2gym_env.py:102-116`, `loaders.py:97,235,307,389,400`).  `load_replay` reads such a file WITHOUT the
reference being importable: a restricted unpickler rebuilds every `ev2gym.*` instance as a plain attribute bag
and refuses any global outside numpy / datetime / builtins containers (a pickle from an untrusted source cannot
run code through it).  What comes out is what the reference itself would simulate from that file -- including
the transformer forecasts as they were left at the end of the recorded episode (the reference overwrites
them in place while it runs, `transformer.py:178-180`, and pickles them in that state).

`objects_to_scenario` does the flattening and also accepts a live reference env's objects;
`replay_tensors` produces the `[ports, chargers, T]` occupancy tensors the reference derives for its
optimal-solver tooling (`replay.py:98-171`) from a `ScenarioBatch`.
"""
from __future__ import annotations

import io
import pickle
from typing import Dict, Optional, Sequence

import numpy as np

from . import _abi
from .scenario import ScenarioBatch, resolve_ports

_SAFE_MODULE_PREFIXES = ("numpy", "datetime", "collections", "copyreg", "_codecs")
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "complex", "bool", "str", "bytes",
                  "bytearray", "object", "slice", "range"}


class _Bag:
    """Attribute bag standing in for an `ev2gym.*` instance (state arrives through `__dict__`)."""

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2:    # (dict, slots) form
            state = {**(state[0] or {}), **(state[1] or {})}
        self.__dict__.update(state or {})


class _ReplayUnpickler(pickle.Unpickler):
    _bags: Dict[str, type] = {}

    def find_class(self, module, name):
        if module == "ev2gym" or module.startswith("ev2gym."):
            key = f"{module}.{name}"
            if key not in self._bags:
                self._bags[key] = type(name, (_Bag,), {"__module__": "ev2gym_amd.replay", "_source": key})
            return self._bags[key]
        if module == "builtins" and name in _SAFE_BUILTINS:
            return super().find_class(module, name)
        if module.split(".")[0] in _SAFE_MODULE_PREFIXES:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"replay file references {module}.{name}: not allowed")


def read_replay_object(path_or_bytes):
    """The unpickled replay as an attribute bag (fields of `EvCityReplay.__init__`, replay.py:23-96)."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    return _ReplayUnpickler(io.BytesIO(data)).load()


def _lut_table(d) -> tuple:
    # 101-entry percent table; the reference looks up `dict.get(round(amps), 1)` (ev.py:287-290)
    if any((k < 0 or k > 100) for k in d.keys()):
        raise ValueError("efficiency table keys outside 0..100 A are not representable")
    return tuple(float(d.get(i, 1)) for i in range(_abi.LUT_LEN))


def objects_to_scenario(transformers: Sequence, charging_stations: Sequence, ev_profiles: Sequence, charge_prices,
                        discharge_prices, power_setpoints, timescale: int, simulation_length: int,
                        v2g_enabled: Optional[bool] = None, horizon: int = 20) -> ScenarioBatch:
    """Flatten reference-shaped objects (attribute names of ev.py:68-113, ev_charger.py:57-94,
    transformer.py:38-78) into a one-env `ScenarioBatch`."""
    T = int(simulation_length)
    cs, trs, evs = list(charging_stations), list(transformers), list(ev_profiles)
    npc = {int(c.n_ports) for c in cs}
    if len(npc) != 1:
        raise NotImplementedError("chargers with different port counts (topology JSON) are out of scope")
    cp, dp = np.asarray(charge_prices, float), np.asarray(discharge_prices, float)
    if cp.ndim == 2:
        if not (cp == cp[0]).all() or not (dp == dp[0]).all():
            raise NotImplementedError("per-charger price rows differ (the loaders never produce this, loaders.py:423-424)")
        cp, dp = cp[0], dp[0]
    rec = {"scn_charge_price": cp[:T], "scn_discharge_price": dp[:T],
           "scn_power_setpoints": np.asarray(power_setpoints, float)[:T]}
    for k, attr, dt in (("cs_min_charge_current", "min_charge_current", float), ("cs_max_charge_current", "max_charge_current", float),
                        ("cs_min_discharge_current", "min_discharge_current", float),
                        ("cs_max_discharge_current", "max_discharge_current", float), ("cs_voltage", "voltage", float),
                        ("cs_phases", "phases", np.int32), ("cs_transformer", "connected_transformer", np.int32)):
        rec["scn_" + k] = np.array([getattr(c, attr) for c in cs], dt)
    for k, attr in (("tr_max_power", "max_power"), ("tr_min_power", "min_power"), ("tr_inflexible_load", "inflexible_load"),
                    ("tr_solar_power", "solar_power"), ("tr_load_forecast", "inflexible_load_forecast"),
                    ("tr_pv_forecast", "pv_generation_forecast")):
        rec["scn_" + k] = np.array([np.asarray(getattr(t, attr), float)[:T] for t in trs], float)
    R = len(trs)
    nd = max([len(t.dr_events) for t in trs] + [1])
    dr, ndr = np.zeros((R, nd, 3)), np.zeros(R, np.int32)
    for i, t in enumerate(trs):
        ndr[i] = len(t.dr_events)
        for j, e in enumerate(t.dr_events):
            dr[i, j] = (e["event_start_step"], e["event_end_step"], e["capacity_percentage"])
    rec.update(scn_tr_dr=dr, scn_tr_n_dr=ndr, scn_tr_steps_ahead=np.array([t.steps_ahead for t in trs], np.int32))
    f = lambda name: np.array([getattr(e, name) for e in evs], float)  # noqa: E731
    rec.update(scn_ev_cs=np.array([e.location for e in evs], np.int32),
               scn_ev_t_arr=np.array([e.time_of_arrival for e in evs], np.int32),
               scn_ev_t_dep=np.array([e.time_of_departure for e in evs], np.int32),
               scn_ev_cap0=f("battery_capacity_at_arrival"), scn_ev_B=f("battery_capacity"),
               scn_ev_desired=f("desired_capacity"), scn_ev_minB=f("min_battery_capacity"),
               scn_ev_min_emerg=f("min_emergency_battery_capacity"), scn_ev_pac_max=f("max_ac_charge_power"),
               scn_ev_pac_min=f("min_ac_charge_power"), scn_ev_pdis_max=f("max_discharge_power"),
               scn_ev_pdis_min=f("min_discharge_power"), scn_ev_ts=f("transition_soc"),
               scn_ev_tsm=f("transition_soc_multiplier"), scn_ev_phases=np.array([e.ev_phases for e in evs], np.int32))
    luts, ids, eta_c, eta_d, lid = [], {}, [], [], []
    for e in evs:
        if isinstance(e.charge_efficiency, dict):
            if e.discharge_efficiency != e.charge_efficiency:
                raise NotImplementedError("different charge / discharge efficiency tables on one EV")
            tab = _lut_table(e.charge_efficiency)
            if tab not in ids:
                ids[tab] = len(luts)
                luts.append(tab)
            lid.append(ids[tab])
            eta_c.append(np.nan)
            eta_d.append(np.nan)
        else:
            lid.append(-1)
            eta_c.append(float(e.charge_efficiency))
            eta_d.append(float(e.discharge_efficiency))
    rec.update(scn_ev_eta_ch=np.array(eta_c, float), scn_ev_eta_dis=np.array(eta_d, float), scn_ev_lut=np.array(lid, np.int32),
               scn_lut=np.array(luts, float).reshape(-1, _abi.LUT_LEN) if luts else np.zeros((0, _abi.LUT_LEN)))
    if v2g_enabled is None:   # the flag lives in the YAML, not in the replay; it only selects the action-space low bound
        v2g_enabled = bool(np.any(rec["scn_cs_max_discharge_current"] != 0))
    rec["scn_meta"] = np.array([T, int(timescale), len(cs), npc.pop(), R, int(bool(v2g_enabled)), int(horizon)], np.int64)
    return ScenarioBatch.from_single(rec)


def load_replay(path_or_bytes, v2g_enabled: Optional[bool] = None) -> ScenarioBatch:
    """`EV2Gym(load_from_replay_path=path)`'s scenario as a one-env `ScenarioBatch` (concat several for a batch)."""
    r = read_replay_object(path_or_bytes)
    if getattr(r, "simulate_grid", False):
        raise NotImplementedError("replays recorded with simulate_grid: True are out of scope (power-flow path)")
    return objects_to_scenario(r.transformers, r.charging_stations, r.EVs, r.charge_prices, r.discharge_prices,
                               r.power_setpoints, r.timescale, r.sim_length, v2g_enabled)


def replay_tensors(batch: ScenarioBatch, env: int = 0) -> Dict[str, np.ndarray]:
    """The scenario-only `[max_n_ports, n_cs, T]` tensors of `EvCityReplay` (replay.py:98-171) for one env:
    `u`, `ev_arrival`, `t_dep`, `energy_at_arrival`, `ev_max_energy`, `ev_max_ch_power`, `ev_max_dis_power`,
    `ev_des_energy`, plus the per-charger limits.  (`max_energy_at_departure` depends on the recorded run, not on
    the scenario, and is not produced.)  Ports are the ones EVs actually occupy (first-free rule)."""
    b = batch.select(np.array([env])) if batch.n_envs > 1 else batch
    a, T, C, npc = b.arrays, b.n_steps, b.n_chargers, b.ports_per_charger
    port = resolve_ports(b)          # [S] port within the charger, -1 = never admitted
    z = lambda: np.zeros((npc, C, T))  # noqa: E731
    out = {k: z() for k in ("u", "ev_arrival", "t_dep", "energy_at_arrival", "ev_max_energy", "ev_max_ch_power",
                            "ev_max_dis_power", "ev_des_energy")}
    for s in range(len(a["ev_cs"])):
        p, c, ta, td0 = int(port[s]) % npc if port[s] >= 0 else -1, int(a["ev_cs"][s]), int(a["ev_t_arr"][s]), int(a["ev_t_dep"][s])
        if p < 0 or ta >= T:
            continue
        td = min(td0, T)
        out["ev_max_energy"][p, c, ta:td] = a["ev_B"][s]
        out["ev_max_ch_power"][p, c, ta:td] = a["ev_pac_max"][s]
        out["ev_max_dis_power"][p, c, ta:td] = a["ev_pdis_max"][s]
        out["u"][p, c, ta:td] = 1
        out["energy_at_arrival"][p, c, ta] = a["ev_cap0"][s]
        out["ev_arrival"][p, c, ta] = 1
        out["t_dep"][p, c, td if td0 < T else td - 1] = 1
        if td < T:
            out["ev_des_energy"][p, c, td] = a["ev_desired"][s]
    out.update(port_max_charge_current=a["cs_max_charge_current"].copy(), port_min_charge_current=a["cs_min_charge_current"].copy(),
               port_max_discharge_current=a["cs_max_discharge_current"].copy(),
               port_min_discharge_current=a["cs_min_discharge_current"].copy(),
               voltages=a["cs_voltage"] * np.sqrt(a["cs_phases"]), cs_transformer=a["cs_transformer"].astype(float),
               charge_prices=np.tile(a["charge_price"][0], (C, 1)), discharge_prices=np.tile(a["discharge_price"][0], (C, 1)),
               power_setpoints=a["power_setpoints"][0].copy())
    return out
