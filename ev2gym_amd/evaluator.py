"""Batched counterpart of the reference's evaluation loop (evaluator.py:102-109, 248-287).

The reference evaluates one (scenario, algorithm) pair at a time: it replays a scenario file, asks the algorithm for an action every
step, and appends the terminal statistics dict to a results table.  Here every scenario of a batch is an env of the engine, an
observation-independent algorithm is a device-resident action source, and a whole evaluation is one fused launch per algorithm
(`ev2g_step_n`, persistent): the table has the reference's columns, one row per (run, algorithm).

Algorithms with a closed-form action source (the reference's rule-based agents that do not look at the observation):
  ChargeAsFastAsPossible (heuristics.py:152-166)  every port at 1
  DoNothing (heuristics.py:533-544)               every port at 0
  RandomAgent (heuristics.py:546-558)             uniform samples of the action box, counter-based generator (seed, index)
Agents that read the env (round robin, charge-as-late-as-possible, MPC, RL policies) go through `EV2GymVec` / the facade step by step.
"""
from __future__ import annotations

import time
from typing import Iterable, Optional

import numpy as np

from . import _abi
from .scenario import ScenarioBatch

ALGORITHMS = ("ChargeAsFastAsPossible", "DoNothing", "RandomAgent")
# the statistics columns of the reference's results table (evaluator.py:262-283), in its order
RESULT_STATS = ["total_ev_served", "total_profits", "total_energy_charged", "total_energy_discharged", "average_user_satisfaction",
                "power_tracker_violation", "tracking_error", "energy_tracking_error", "energy_user_satisfaction", "total_transformer_overload",
                "battery_degradation", "battery_degradation_calendar", "battery_degradation_cycling"]


def _default_engine(batch, rk, sk):
    from .engine import Engine   # the HIP engine; fails loudly without a GPU
    return Engine(batch, rk, sk, flags=_abi.FLAG_LOG_SOC)


def evaluate(scenarios: ScenarioBatch, algorithms: Iterable[str] = ALGORITHMS, state_function="V2G_profit_max_loads",
             reward_function="ProfitMax_TrPenalty_UserIncentives", seed: int = 0, discharge_price_factor: Optional[float] = None,
             engine_factory=_default_engine):
    """Runs every scenario of `scenarios` (e.g. `load_replay` files concatenated, or a generated batch) under every algorithm and
    returns a pandas DataFrame with the reference's columns: run, Algorithm, control_horizon, discharge_price_factor, the thirteen
    statistics of its table, total_reward, time (seconds of GPU kernel time for the algorithm's whole batch, shared by its rows)."""
    import pandas as pd
    sk = _abi.STATE_KINDS[state_function if isinstance(state_function, str) else state_function.__name__]
    rk = _abi.REWARD_KINDS[reward_function if isinstance(reward_function, str) else reward_function.__name__]
    E, T, P = scenarios.n_envs, scenarios.n_steps, scenarios.n_ports
    lo = -1.0 if scenarios.v2g_enabled else 0.0
    rows = []
    for name in algorithms:
        if name not in ALGORITHMS:
            raise NotImplementedError(f"evaluate(): '{name}' reads the env; closed-form action sources are {ALGORITHMS}")
        eng = engine_factory(scenarios, rk, sk)
        try:
            if name == "RandomAgent":
                acts, stride = eng.empty((T, E, P)), E * P
                eng.fill_uniform(acts, T * E * P, seed, lo, 1.0)
            else:
                acts, stride = eng.empty((E, P)), 0      # one [E,P] block reused every step
                acts.upload(np.full((E, P), 1.0 if name == "ChargeAsFastAsPossible" else 0.0))
            eng.reset()
            t0 = time.perf_counter()
            eng.step_n(T, acts, stride, auto_reset=0, persistent=True)
            st = eng.stats()
            wall = time.perf_counter() - t0
            eng.check_faults()
            kernel_s = eng.last_step_n_kernel_ms() / 1e3
        finally:
            eng.close()
        idx = {n: i for i, n in enumerate(_abi.STAT_NAMES)}
        for run in range(E):
            row = {"run": run, "Algorithm": name, "control_horizon": 0,
                   "discharge_price_factor": discharge_price_factor if discharge_price_factor is not None else float("nan")}
            row.update({k: float(st[run, idx[k]]) for k in RESULT_STATS})
            row["total_reward"] = float(st[run, idx["total_reward"]])
            row["time"] = kernel_s if kernel_s > 0 else wall
            rows.append(row)
    return pd.DataFrame(rows)
