#!/usr/bin/env python3
"""How much does the general kernel gain from a second resident workgroup per CU?  (development probe)
Same total number of ports, two shapes: 2048 envs x 1000 chargers / 50 transformers (ev2g_step_v2<1024>, 148 KB of LDS: one
workgroup per CU) and 4096 envs x 500 chargers / 25 transformers (ev2g_step_v2<512>, 74 KB: two per CU)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2gym_amd import _abi
from ev2gym_amd.engine import Engine
from ev2gym_amd.scenario_gen import GenConfig, generate
for C_, R_, E in ((1000, 50, 2048), (500, 25, 4096), (250, 12, 8192)):
    b = generate(GenConfig.v2g_profit_plus_loads(E, C_, R_, seed=0))
    eng = Engine(b, _abi.REWARD_KINDS["ProfitMax_TrPenalty_UserIncentives"], _abi.STATE_KINDS["V2G_profit_max_loads"], flags=_abi.FLAG_LOG_SOC)
    P, D, T = eng.P, eng.D, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, -1.0, 1.0)
    obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    for persistent in (True, False):
        ms = []
        for rep in range(3):
            eng.reset(obs)
            eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=persistent)
            eng.synchronize()
            ms.append(eng.last_step_n_kernel_ms())
        us = min(ms) * 1e3 / T
        print(f"{eng.kernel_name:22s} E={E:5d} P={P:5d} R={R_:3d} persistent={persistent}: {us:8.2f} us/step  {E*P/us/1e3:8.2f} G port-steps/s")
    eng.close()
