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

# exact (module, name) pairs a replay pickle may reference: the reference's four replay classes (as attribute bags), numpy's
# array / scalar / dtype reconstruction helpers under both module spellings (numpy 1.x "numpy.core", 2.x "numpy._core"),
# datetime for sim_date, and plain containers.  Anything else -- any other numpy or ev2gym name included -- is refused.
_REPLAY_CLASSES = {("ev2gym.models.replay", "EvCityReplay"), ("ev2gym.models.ev", "EV"), ("ev2gym.models.ev_charger", "EV_Charger"),
                   ("ev2gym.models.transformer", "Transformer")}
_SAFE_GLOBALS = {("numpy", "ndarray"), ("numpy", "dtype"), ("datetime", "datetime"), ("datetime", "date"), ("datetime", "timedelta"),
                 ("collections", "OrderedDict"), ("_codecs", "encode"), ("copyreg", "_reconstructor")}
_SAFE_GLOBALS |= {(m + sub, n) for m in ("numpy.core", "numpy._core") for sub, n in
                  ((".multiarray", "_reconstruct"), (".multiarray", "scalar"), (".numeric", "_frombuffer"))}
_SAFE_GLOBALS |= {("builtins", n) for n in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "complex", "bool", "str",
                                            "bytes", "bytearray", "object", "slice", "range")}


class _Bag:
    """Attribute bag standing in for an `ev2gym.*` instance (state arrives through `__dict__`)."""

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2:    # (dict, slots) form
            state = {**(state[0] or {}), **(state[1] or {})}
        self.__dict__.update(state or {})


class _ReplayUnpickler(pickle.Unpickler):
    _bags: Dict[str, type] = {}

    def find_class(self, module, name):
        if (module, name) in _REPLAY_CLASSES:
            key = f"{module}.{name}"
            if key not in self._bags:
                self._bags[key] = type(name, (_Bag,), {"__module__": "ev2gym_amd.replay", "_source": key})
            return self._bags[key]
        if (module, name) in _SAFE_GLOBALS:
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
    n_ports = np.array([int(c.n_ports) for c in cs], np.int32)
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
    rec["scn_cs_n_ports"] = n_ports
    rec["scn_meta"] = np.array([T, int(timescale), len(cs), int(n_ports.max()), R, int(bool(v2g_enabled)), int(horizon)], np.int64)
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
    port = resolve_ports(b) - b.port_base[a["ev_cs"]]          # [S] port within the charger
    z = lambda: np.zeros((npc, C, T))  # noqa: E731
    out = {k: z() for k in ("u", "ev_arrival", "t_dep", "energy_at_arrival", "ev_max_energy", "ev_max_ch_power",
                            "ev_max_dis_power", "ev_des_energy")}
    for s in range(len(a["ev_cs"])):
        p, c, ta, td0 = int(port[s]), int(a["ev_cs"][s]), int(a["ev_t_arr"][s]), int(a["ev_t_dep"][s])
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


# ---- writing -----------------------------------------------------------------------------------------------------------
# The reference's loaders take a replay as a pickled object graph of ITS classes (ev2gym_env.py:102-116, loaders.py:97,236,
# 308,389,401): EvCityReplay holding Transformer / EV_Charger / EV instances.  A pickle names a class by (module, qualname)
# and stores the instance __dict__, so such a file can be written without importing the reference: stand-in classes carry
# the reference's (module, name), and a pickler that emits exactly that GLOBAL for them.

def _ref_class(module: str, name: str):
    return type(name, (object,), {"_ev2g_ref_global": (module, name), "__module__": module})


_REF = {k: _ref_class(m, k) for k, m in (("EvCityReplay", "ev2gym.models.replay"), ("EV", "ev2gym.models.ev"),
                                          ("EV_Charger", "ev2gym.models.ev_charger"), ("Transformer", "ev2gym.models.transformer"))}


class _ReplayPickler(pickle._Pickler):   # the pure-Python pickler: how a class is named can be overridden
    def save(self, obj, save_persistent_id=True):
        # intercepted in save(), ahead of the type dispatch table: packages such as dill patch that table process-wide
        ref = obj.__dict__.get("_ev2g_ref_global") if isinstance(obj, type) else None
        if ref is None:
            return super().save(obj, save_persistent_id)
        seen = self.memo.get(id(obj))
        if seen is not None:
            return self.write(self.get(seen[0]))
        self.write(pickle.GLOBAL + ref[0].encode("ascii") + b"\n" + ref[1].encode("ascii") + b"\n")
        self.memoize(obj)


def _new(kind: str, **attrs):
    o = _REF[kind]()
    o.__dict__.update(attrs)
    return o


def replay_objects(batch: ScenarioBatch, env: int = 0, run: Optional[dict] = None, stats: Optional[dict] = None,
                   sim_date=None, scenario: str = "workplace", heterogeneous_specs: bool = True, sim_name: Optional[str] = None,
                   replay_path: str = "./replay/"):
    """The `EvCityReplay` object graph (attribute for attribute: replay.py:23-171, ev.py:68-136, ev_charger.py:57-112,
    transformer.py:38-78) of one env of a scenario batch.  `run` carries what an episode left behind (Engine.peek of the
    finished env + `stats`); without it the run-dependent fields hold their reset values."""
    import datetime
    b = batch.select(np.array([env])) if batch.n_envs > 1 else batch
    a, T, C, npc, R, dt = b.arrays, b.n_steps, b.n_chargers, b.ports_per_charger, b.n_transformers, b.timescale
    run = run or {}
    port = resolve_ports(b) - b.port_base[a["ev_cs"]]     # port within the charger
    sim_date = sim_date or datetime.datetime(2022, 1, 1, 5, 0)
    sim_name = sim_name or ("sim_" + sim_date.strftime("%Y_%m_%d") + "_000000")
    volt_cfg = float(a["cs_voltage"][0]) * float(np.sqrt(a["cs_phases"][0]))   # Transformer.voltage: config voltage * sqrt(phases) (transformer.py:39-40)
    trs = []
    for r in range(R):
        maxp, minp = a["tr_max_power"][0, r].copy(), a["tr_min_power"][0, r].copy()
        dr = [dict(event_start_step=int(e[0]), event_end_step=int(e[1]), capacity_percentage=float(e[2]))
              for e in a["tr_dr"][0, r][:int(a["tr_n_dr"][0, r])]]
        trs.append(_new("Transformer", id=r, voltage=volt_cfg, max_current=maxp * 1000 / volt_cfg, min_current=minp * 1000 / volt_cfg,
                        max_power=maxp, min_power=minp, inflexible_load=a["tr_inflexible_load"][0, r].copy(),
                        solar_power=a["tr_solar_power"][0, r].copy(), cs_ids=np.where(a["cs_transformer"] == r)[0].astype(np.int64),
                        simulation_length=T, current_amps=float(run.get("tr_power", np.zeros(R))[r]) * 1000 / volt_cfg,
                        current_power=float(run.get("tr_power", np.zeros(R))[r]), current_step=max(int(run.get("current_step", 0)) - 1, 0),
                        inflexible_load_forecast=a["tr_load_forecast"][0, r].copy(), pv_generation_forecast=a["tr_pv_forecast"][0, r].copy(),
                        steps_ahead=int(a["tr_steps_ahead"][0, r]), dr_events=dr))
    zC = np.zeros(C)
    css = []
    for c in range(C):
        css.append(_new("EV_Charger", id=c, connected_bus=0, connected_transformer=int(a["cs_transformer"][c]), geo_location=None,
                        n_ports=int(a["cs_n_ports"][c]), charger_type="AC", timescale=dt, min_charge_current=float(a["cs_min_charge_current"][c]),
                        max_charge_current=float(a["cs_max_charge_current"][c]), min_discharge_current=float(a["cs_min_discharge_current"][c]),
                        max_discharge_current=float(a["cs_max_discharge_current"][c]), phases=int(a["cs_phases"][c]),
                        voltage=float(a["cs_voltage"][c]), current_power_output=0, evs_connected=[None] * int(a["cs_n_ports"][c]), n_evs_connected=0,
                        current_step=int(run.get("current_step", 0)), current_charge_price=0, current_discharge_price=0,
                        current_total_amps=0, current_signal=[], total_energy_charged=float(run.get("cs_energy_charged", zC)[c]),
                        total_energy_discharged=float(run.get("cs_energy_discharged", zC)[c]), total_profits=float(run.get("cs_profits", zC)[c]),
                        total_evs_served=0, total_user_satisfaction=0, all_user_satisfaction=[], verbose=False))
    lut = a["lut"]
    evs = []
    for s in range(len(a["ev_cs"])):
        lid = int(a["ev_lut"][s])
        eff_c = {i: float(lut[lid, i]) for i in range(_abi.LUT_LEN)} if lid >= 0 else float(a["ev_eta_ch"][s])
        eff_d = dict(eff_c) if lid >= 0 else float(a["ev_eta_dis"][s])
        cap0, B = float(a["ev_cap0"][s]), float(a["ev_B"][s])
        evs.append(_new("EV", id=int(port[s]), location=int(a["ev_cs"][s]), timescale=dt,
                        time_of_arrival=int(a["ev_t_arr"][s]), time_of_departure=int(a["ev_t_dep"][s]), desired_capacity=float(a["ev_desired"][s]),
                        battery_capacity_at_arrival=cap0, battery_capacity=B, min_battery_capacity=float(a["ev_minB"][s]),
                        min_emergency_battery_capacity=float(a["ev_min_emerg"][s]), max_ac_charge_power=float(a["ev_pac_max"][s]),
                        min_ac_charge_power=float(a["ev_pac_min"][s]), max_discharge_power=float(a["ev_pdis_max"][s]),
                        min_discharge_power=float(a["ev_pdis_min"][s]), max_dc_charge_power=50, transition_soc=float(a["ev_ts"][s]),
                        transition_soc_multiplier=float(a["ev_tsm"][s]), ev_phases=int(a["ev_phases"][s]), charge_efficiency=eff_c,
                        discharge_efficiency=eff_d, current_capacity=cap0, prev_capacity=cap0, current_energy=0, actual_current=0,
                        charging_cycles=0, previous_power=0, required_energy=B - cap0, total_energy_exchanged=0, max_energy_AFAP=0,
                        min_emergency_battery_capacity_metric=0, abs_total_energy_exchanged=0, historic_soc=[], active_steps=[],
                        calendar_loss=0, cyclic_loss=0))
    t = replay_tensors(b)
    usage = np.asarray(run.get("power_usage", np.zeros(T)), float)
    rep = _new("EvCityReplay", stats=dict(stats or {}), replay_path=replay_path + "replay_" + sim_name + ".pkl", sim_name=sim_name + "_replay",
               sim_length=T, n_cs=C, n_transformers=R, timescale=dt, sim_date=sim_date, cs_transformers=[int(x) for x in a["cs_transformer"]],
               power_setpoints=a["power_setpoints"][0].copy(), scenario=scenario, heterogeneous_specs=bool(heterogeneous_specs),
               ev_load_potential=usage, simulate_grid=False, transformers=trs, charging_stations=css, EVs=evs, grid=None,
               unstirred_EVs=None, unstirred_stats=None, optimal_EVs=None, optimal_stats=None,
               charge_prices=t["charge_prices"], discharge_prices=t["discharge_prices"])
    infl, sol = a["tr_inflexible_load"][0], a["tr_solar_power"][0]
    rep.tra_max_amps = np.stack([trs[r].max_current - np.abs(infl[r] * 1000 / volt_cfg) + np.abs(sol[r] * 1000 / volt_cfg) for r in range(R)])
    rep.tra_min_amps = np.stack([trs[r].min_current + np.abs(infl[r] * 1000 / volt_cfg) - np.abs(sol[r] * 1000 / volt_cfg) for r in range(R)])
    rep.port_max_charge_current, rep.port_min_charge_current = t["port_max_charge_current"], t["port_min_charge_current"]
    rep.port_max_discharge_current, rep.port_min_discharge_current = t["port_max_discharge_current"], t["port_min_discharge_current"]
    rep.voltages, rep.phases = t["voltages"], np.ones(C)
    rep.cs_ch_efficiency, rep.cs_dis_efficiency = np.ones((C, T)), np.ones((C, T))
    rep.cs_transformer = t["cs_transformer"]
    rep.max_n_ports = npc
    rep.ev_max_energy, rep.ev_min_energy = t["ev_max_energy"], np.zeros((npc, C, T))
    rep.ev_max_ch_power, rep.ev_max_dis_power = t["ev_max_ch_power"], t["ev_max_dis_power"]
    rep.u, rep.energy_at_arrival, rep.ev_arrival, rep.t_dep, rep.ev_des_energy = t["u"], t["energy_at_arrival"], t["ev_arrival"], t["t_dep"], t["ev_des_energy"]
    # max_energy_at_departure (replay.py:158-169) is the capacity BEFORE an EV's last charging step, capped at the battery size: a
    # quantity of the recorded run that only the optimal-solver tooling reads.  From a finished run: the final capacities, capped.
    mead = np.zeros((npc, C, T))
    fin = run.get("session_final_cap")
    if fin is not None:
        for s in range(len(a["ev_cs"])):
            td0, ta = int(a["ev_t_dep"][s]), int(a["ev_t_arr"][s])
            if ta >= T or not np.isfinite(fin[s]):
                continue
            mead[int(port[s]), int(a["ev_cs"][s]), td0 if td0 < T else T - 1] = min(float(fin[s]), float(a["ev_B"][s]))
    rep.max_energy_at_departure = mead
    return rep


def write_replay(path: str, batch: ScenarioBatch, env: int = 0, **kw) -> str:
    """Write one env of a scenario batch as a replay file the reference's `EV2Gym(load_from_replay_path=...)` accepts
    (a pickle of its EvCityReplay object graph, models/replay.py:10-174; ev2gym_env.py:503-510 is the reference's writer).
    Keyword arguments as for `replay_objects`."""
    rep = replay_objects(batch, env, **kw)
    with open(path, "wb") as f:
        _ReplayPickler(f, protocol=4).dump(rep)
    return path
