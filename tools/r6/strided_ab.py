#!/usr/bin/env python3
"""cfg2 persistent launch: specialisation 2 (outputs in place), specialisation 3 with only the reward rows kept (same code as the strided launch, almost no
extra bytes), specialisation 3 with every row kept -- is the +19 % the code or the bytes?  (development tool)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ev2gym_amd import engine, _abi
from bench import WORKLOADS
from ev2gym_amd.scenario_gen import generate
wl = WORKLOADS["cfg2"]
E = wl["envs"]
eng = engine.Engine(generate(wl["gen"](E, 0)), _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=_abi.FLAG_LOG_SOC)
P, D, T = eng.P, eng.D, eng.T
acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
obs, rew, done, mask = eng.empty((T, E, D)), eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
modes = {"in place (0,0,0,0)": (0, 0, 0, 0), "reward rows kept only": (0, E, 0, 0), "reward+done+mask kept": (0, E, E, E * P), "obs kept only": (E * D, 0, 0, 0), "every row kept": (E * D, E, E, E * P)}
for rep in range(2):
    for name, s in modes.items():
        ts = []
        for _ in range(8):
            eng.reset(obs)
            eng.step_n(T, acts, E * P, obs, s[0], rew, s[1], done, s[2], mask, s[3], auto_reset=False, persistent=True)
            eng.synchronize()
            ts.append(eng.last_step_n_kernel_ms() * 1e3 / T)
        print(f"{name:28s} specialisation {eng.last_launch_specialisation}: {np.median(ts):.3f} us/step")
