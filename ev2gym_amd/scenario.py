"""Scenario batches: the tensors `EV2Gym.step()` reads, for E independent envs.

This is the host-side data format at the drop-in boundary (include/ev2g.h `ev2g_scenario_batch`).
In the reference these tensors are built by `EV2Gym.__init__/reset()` (ev2gym_env.py:38-331) from
the YAML config through `load_ev_charger_profiles`, `load_transformers`, `EV_spawner`,
`load_electricity_prices`, `load_power_setpoints` (utilities/loaders.py, utilities/utils.py); here
they are plain numpy arrays so that thousands of envs can be packed into HBM at once.

Field reference (E envs, C chargers, R transformers, T steps, S sessions total):
  cs_*            [C]      charger statics (EV_Charger.__init__, ev_charger.py:41-94)
  charge_price    [E,T]    row 0 of env.charge_prices (negative EUR/kWh; loaders.py:439-442)
  discharge_price [E,T]
  power_setpoints [E,T]
  tr_*            [E,R,T]  Transformer arrays (transformer.py:38-78); the forecasts as they stand
                           after reset(); tr_dr [E,R,ND,3] demand-response events
  env_session_start [E+1]  CSR offsets into the session arrays
  ev_*            [S]      one row per EV session in EVs_profiles (arrival) order
  lut             [NL,101] charge-efficiency tables in percent (utils.py:273-290)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _abi

_INT_SCALARS = ["n_envs", "n_steps", "timescale", "n_chargers", "ports_per_charger", "n_transformers",
                "horizon", "n_dr_max", "n_lut"]
_NP = {C.c_double: np.float64, C.c_int32: np.int32, C.c_int64: np.int64}


@dataclass
class ScenarioBatch:
    n_envs: int
    n_steps: int
    timescale: int
    n_chargers: int
    ports_per_charger: int
    n_transformers: int
    v2g_enabled: bool = True
    horizon: int = 20
    arrays: Dict[str, np.ndarray] = field(default_factory=dict)

    # ---- derived sizes -----------------------------------------------------------------
    @property
    def n_ports(self) -> int:
        """Ports of one env: sum of the chargers' n_ports (ev2gym_env.py:201-202)."""
        n = self.arrays.get("cs_n_ports")
        return int(np.sum(n)) if n is not None else self.n_chargers * self.ports_per_charger

    @property
    def port_base(self) -> np.ndarray:
        """[C+1] first port of every charger in the cumulative numbering (ev2gym_env.py:364-385)."""
        n = self.arrays.get("cs_n_ports")
        n = np.full(self.n_chargers, self.ports_per_charger) if n is None else n
        return np.concatenate([[0], np.cumsum(n)]).astype(np.int64)

    @property
    def uniform_ports(self) -> bool:
        n = self.arrays.get("cs_n_ports")
        return n is None or bool((np.asarray(n) == self.ports_per_charger).all())

    @property
    def n_sessions(self) -> int:
        return int(self.arrays["env_session_start"][-1])

    @property
    def n_lut(self) -> int:
        return int(self.arrays["lut"].shape[0])

    @property
    def n_dr_max(self) -> int:
        return int(self.arrays["tr_dr"].shape[2])

    def obs_dim(self, state_kind: int) -> int:
        P, R = self.n_ports, self.n_transformers
        if state_kind == _abi.STATE_KINDS["PublicPST"]:
            return 3 + 3 * P
        if state_kind == _abi.STATE_KINDS["V2G_profit_max"]:
            return 2 + self.horizon + 2 * P
        return 2 + self.horizon + 2 * self.horizon * R + 2 * P

    def __getattr__(self, name):
        arrays = self.__dict__.get("arrays", {})
        if name in arrays:
            return arrays[name]
        raise AttributeError(name)

    # ---- validation / normalisation ------------------------------------------------------
    def finalize(self) -> "ScenarioBatch":
        E, T, Cn, R = self.n_envs, self.n_steps, self.n_chargers, self.n_transformers
        a = self.arrays
        if "cs_n_ports" not in a:      # no topology file: every charger has ports_per_charger ports
            a["cs_n_ports"] = np.full(Cn, self.ports_per_charger, np.int32)
        for name, ct in _abi.BATCH_ARRAYS:
            if name not in a:
                raise ValueError(f"scenario batch lacks '{name}'")
            a[name] = np.ascontiguousarray(a[name], dtype=_NP[ct])
        S = self.n_sessions
        shapes = {"charge_price": (E, T), "discharge_price": (E, T), "power_setpoints": (E, T),
                  "tr_n_dr": (E, R), "tr_steps_ahead": (E, R), "env_session_start": (E + 1,),
                  "tr_dr": (E, R, a["tr_dr"].shape[2] if a["tr_dr"].ndim == 4 else -1, 3)}
        for n in ("tr_max_power", "tr_min_power", "tr_inflexible_load", "tr_solar_power",
                  "tr_load_forecast", "tr_pv_forecast"):
            shapes[n] = (E, R, T)
        for n, _ in _abi.BATCH_ARRAYS:
            if n.startswith("cs_"):
                shapes[n] = (Cn,)
            elif n.startswith("ev_"):
                shapes[n] = (S,)
        for n, shp in shapes.items():
            if tuple(a[n].shape) != tuple(shp):
                raise ValueError(f"{n}: shape {a[n].shape}, expected {shp}")
        if a["lut"].ndim != 2 or a["lut"].shape[1] != _abi.LUT_LEN:
            raise ValueError("lut must be [NL,101]")
        if S and (a["ev_lut"].max() >= self.n_lut):
            raise ValueError("ev_lut id out of range")
        if (a["cs_transformer"] < 0).any() or (a["cs_transformer"] >= R).any():
            raise ValueError("cs_transformer out of range")
        if self.horizon != 20:
            raise ValueError("horizon must be 20 (state.py:119,129-132)")
        if (a["cs_n_ports"] < 1).any() or int(a["cs_n_ports"].max()) != self.ports_per_charger:
            raise ValueError("cs_n_ports must be >= 1 with ports_per_charger as their maximum")
        return self

    # ---- C view ----------------------------------------------------------------------------
    def to_c(self) -> _abi.ScenarioBatchC:
        """ctypes struct borrowing this object's numpy buffers (keep `self` alive while in use)."""
        self.finalize()
        s = _abi.ScenarioBatchC()
        for n in _INT_SCALARS:
            setattr(s, n, int(getattr(self, n)))
        s.n_sessions = self.n_sessions
        for name, ct in _abi.BATCH_ARRAYS:
            setattr(s, name, self.arrays[name].ctypes.data_as(C.POINTER(ct)))
        return s

    # ---- constructors ------------------------------------------------------------------------
    @staticmethod
    def from_single(rec: Dict[str, np.ndarray]) -> "ScenarioBatch":
        """One-env batch from a golden fixture / single-env scenario record (keys `scn_*`)."""
        g = lambda k: np.asarray(rec["scn_" + k])  # noqa: E731
        T, ts, Cn, npc, R, v2g, H = [int(x) for x in g("meta")]
        a = {}
        for n, _ in _abi.BATCH_ARRAYS:
            if n.startswith("cs_") and ("scn_" + n) in rec:
                a[n] = g(n)
        a["charge_price"] = g("charge_price")[None]
        a["discharge_price"] = g("discharge_price")[None]
        a["power_setpoints"] = g("power_setpoints")[None]
        for n in ("tr_max_power", "tr_min_power", "tr_inflexible_load", "tr_solar_power",
                  "tr_load_forecast", "tr_pv_forecast", "tr_dr", "tr_n_dr", "tr_steps_ahead"):
            a[n] = g(n)[None]
        S = len(g("ev_cs"))
        a["env_session_start"] = np.array([0, S], np.int64)
        for n, _ in _abi.BATCH_ARRAYS:
            if n.startswith("ev_"):
                a[n] = g(n)
        a["lut"] = g("lut").reshape(-1, _abi.LUT_LEN)
        return ScenarioBatch(1, T, ts, Cn, npc, R, bool(v2g), H, a).finalize()

    @staticmethod
    def concat(batches: Sequence["ScenarioBatch"]) -> "ScenarioBatch":
        """Stack batches that share one config (same sizes and charger statics) along the env axis."""
        b0 = batches[0]
        for b in batches[1:]:
            for n in ("n_steps", "timescale", "n_chargers", "ports_per_charger", "n_transformers"):
                if getattr(b, n) != getattr(b0, n):
                    raise ValueError(f"cannot concat: {n} differs")
            for n, _ in _abi.BATCH_ARRAYS:
                if n.startswith("cs_") and not np.array_equal(b.arrays[n], b0.arrays[n]):
                    raise ValueError(f"cannot concat: {n} differs")
        a = {n: b0.arrays[n] for n, _ in _abi.BATCH_ARRAYS if n.startswith("cs_")}
        nd = max(b.n_dr_max for b in batches)

        def pad_dr(x):
            if x.shape[2] == nd:
                return x
            out = np.zeros(x.shape[:2] + (nd, 3))
            out[:, :, :x.shape[2]] = x
            return out
        for n in ("charge_price", "discharge_price", "power_setpoints", "tr_max_power", "tr_min_power",
                  "tr_inflexible_load", "tr_solar_power", "tr_load_forecast", "tr_pv_forecast",
                  "tr_n_dr", "tr_steps_ahead"):
            a[n] = np.concatenate([b.arrays[n] for b in batches], 0)
        a["tr_dr"] = np.concatenate([pad_dr(b.arrays["tr_dr"]) for b in batches], 0)
        # efficiency tables: concatenate and re-base ids
        luts, lut_off, off = [], [], 0
        for b in batches:
            lut_off.append(off)
            luts.append(b.arrays["lut"])
            off += b.n_lut
        a["lut"] = np.concatenate(luts, 0) if off else np.zeros((0, _abi.LUT_LEN))
        starts = [np.zeros(1, np.int64)]
        base = 0
        for b in batches:
            starts.append(b.arrays["env_session_start"][1:] + base)
            base += b.n_sessions
        a["env_session_start"] = np.concatenate(starts)
        for n, _ in _abi.BATCH_ARRAYS:
            if n.startswith("ev_") and n != "ev_lut":
                a[n] = np.concatenate([b.arrays[n] for b in batches])
        a["ev_lut"] = np.concatenate([np.where(b.arrays["ev_lut"] >= 0, b.arrays["ev_lut"] + o, -1)
                                      for b, o in zip(batches, lut_off)]).astype(np.int32)
        out = ScenarioBatch(sum(b.n_envs for b in batches), b0.n_steps, b0.timescale, b0.n_chargers,
                            b0.ports_per_charger, b0.n_transformers, b0.v2g_enabled, b0.horizon, a)
        return out.dedup_luts().finalize()

    def dedup_luts(self) -> "ScenarioBatch":
        lut = self.arrays["lut"]
        if len(lut) == 0:
            return self
        uniq, inv = np.unique(lut, axis=0, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
        ids = self.arrays["ev_lut"]
        self.arrays["ev_lut"] = np.where(ids >= 0, inv[np.maximum(ids, 0)], -1).astype(np.int32)
        self.arrays["lut"] = uniq
        return self

    def select(self, env_ids: Sequence[int]) -> "ScenarioBatch":
        """Sub-batch with the given envs (used for sharding across GPUs and for tiling pools)."""
        env_ids = np.asarray(env_ids, np.int64)
        a = {n: self.arrays[n] for n, _ in _abi.BATCH_ARRAYS if n.startswith("cs_")}
        for n in ("charge_price", "discharge_price", "power_setpoints", "tr_max_power", "tr_min_power",
                  "tr_inflexible_load", "tr_solar_power", "tr_load_forecast", "tr_pv_forecast", "tr_dr",
                  "tr_n_dr", "tr_steps_ahead"):
            a[n] = self.arrays[n][env_ids]
        st = self.arrays["env_session_start"]
        cnt = (st[1:] - st[:-1])[env_ids]
        a["env_session_start"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        idx = np.concatenate([np.arange(st[e], st[e + 1]) for e in env_ids]) if len(env_ids) else np.zeros(0, np.int64)
        idx = idx.astype(np.int64)
        for n, _ in _abi.BATCH_ARRAYS:
            if n.startswith("ev_"):
                a[n] = self.arrays[n][idx]
        a["lut"] = self.arrays["lut"]
        return ScenarioBatch(len(env_ids), self.n_steps, self.timescale, self.n_chargers, self.ports_per_charger,
                             self.n_transformers, self.v2g_enabled, self.horizon, a).finalize()

    def busy_window(self):
        """(first, last) step of every scenario in which a port holds an EV: first arrival .. last departure (T, -1 for a scenario without
        EVs).  Occupancy does not depend on the actions (ev.py:191-202), so this is a property of the scenario."""
        st = self.arrays["env_session_start"]
        ta, td = self.arrays["ev_t_arr"], self.arrays["ev_t_dep"]
        first = np.full(self.n_envs, self.n_steps, np.int64)
        last = np.full(self.n_envs, -1, np.int64)
        env_of = np.repeat(np.arange(self.n_envs), np.diff(st))
        if len(env_of):
            np.minimum.at(first, env_of, ta)
            np.maximum.at(last, env_of, np.minimum(td, self.n_steps - 1))
        return first, last

    def sorted_by_busy_window(self, window: Optional[int] = None) -> "ScenarioBatch":
        """The same scenarios, re-ordered so that neighbours have similar busy windows (within consecutive blocks of `window` scenarios, or
        over the whole batch).  The step kernel advances a few envs per workgroup in lockstep; a workgroup whose envs are all idle skips the
        battery-maths phase of that step, so co-scheduling envs that wake up and fall idle together saves whole phases (an independent,
        identically distributed pool of generated scenarios has no meaningful order; a batch whose order matters -- replay files -- should not
        be sorted)."""
        first, last = self.busy_window()
        n = self.n_envs
        w = n if not window else int(window)
        order = np.concatenate([b0 + np.lexsort((last[b0:b0 + w], first[b0:b0 + w])) for b0 in range(0, n, w)])
        return self.select(order)

    def tile(self, n_envs: int) -> "ScenarioBatch":
        """Repeat this batch's envs cyclically up to n_envs (scenario pool reuse)."""
        return self.select(np.arange(n_envs) % self.n_envs)

    def shard(self, rank: int, world: int) -> "ScenarioBatch":
        """Contiguous env range of `rank` out of `world`."""
        lo = rank * self.n_envs // world
        hi = (rank + 1) * self.n_envs // world
        return self.select(np.arange(lo, hi))

    # ---- disk ------------------------------------------------------------------------------
    def save(self, path: str) -> None:
        meta = np.array([self.n_envs, self.n_steps, self.timescale, self.n_chargers, self.ports_per_charger,
                         self.n_transformers, int(self.v2g_enabled), self.horizon], np.int64)
        np.savez_compressed(path, batch_meta=meta, **self.arrays)

    @staticmethod
    def load(path: str) -> "ScenarioBatch":
        z = np.load(path)
        m = [int(x) for x in z["batch_meta"]]
        a = {n: z[n] for n, _ in _abi.BATCH_ARRAYS if n in z}
        return ScenarioBatch(m[0], m[1], m[2], m[3], m[4], m[5], bool(m[6]), m[7], a).finalize()


def resolve_ports(batch: ScenarioBatch) -> np.ndarray:
    """Port index (first port of the charger + slot) of every session, by replaying
    EV_Charger.spawn_ev's first-free rule (ev_charger.py:266-286).  Occupancy does not depend on the
    actions: an EV attached at the end of step t_arr-1 leaves at the end of step t_dep
    (ev_charger.py:209-229, ev.py:191-202), and departures of a step precede its arrivals
    (ev2gym_env.py:363-417).  Host-side mirror of what ev2g_load_scenarios() does natively."""
    a = batch.arrays
    npc, base = batch.ports_per_charger, batch.port_base
    n_of = np.diff(base)
    out = np.full(batch.n_sessions, -1, np.int32)
    st = a["env_session_start"]
    for e in range(batch.n_envs):
        free_at = np.zeros((batch.n_chargers, npc), np.int64)  # first step index at which the slot is free again
        for s in range(st[e], st[e + 1]):
            cs, ta, td = int(a["ev_cs"][s]), int(a["ev_t_arr"][s]), int(a["ev_t_dep"][s])
            # attached at end of step ta-1; slot busy through step td (freed inside step td before spawns)
            slot = -1
            for j in range(int(n_of[cs])):
                if free_at[cs, j] <= ta - 1:
                    slot = j
                    break
            if slot < 0:
                raise ValueError(f"env {e}: no free port on charger {cs} at step {ta}")
            free_at[cs, slot] = td
            out[s] = base[cs] + slot
    return out
