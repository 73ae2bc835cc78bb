#!/usr/bin/env python3
"""Profiling driver (development): N persistent whole-episode launches of one workload, nothing else.  python tools/r6/run_cfg.py cfg4 [launches]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import WORKLOADS
from ev2gym_amd import _abi
from ev2gym_amd.engine import Engine
from ev2gym_amd.scenario_gen import generate_native
w = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = WORKLOADS[w]
E = int(os.environ.get("AB_ENVS", wl["envs"]))
batch = generate_native(wl["gen"](2 * E, 0)).sorted_by_busy_window(E)
eng = Engine(batch, _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], device=0, flags=_abi.FLAG_LOG_SOC, n_active_envs=E)
P, D, T = eng.P, eng.D, eng.T
acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
ms = []
for r in range(n):
    eng.reset(obs, offset=(r % 2) * E)
    eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
    ms.append(eng.last_step_n_kernel_ms())
eng.check_faults()
print(f"{w}: {eng.kernel_name} spec {eng.last_launch_specialisation}: {np.median(ms) * 1e3 / T:.3f} us/step (median of {n} launches)")
