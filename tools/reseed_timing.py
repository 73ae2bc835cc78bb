#!/usr/bin/env python3
"""How long does `EV2GymVec.reset(seed=new)` take (host scenario generation + ev2g_load_scenarios)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2gym_amd.vec_env import EV2GymVec
cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ev2gym_amd", "example_config_files", "V2GProfitPlusLoads_50cs.yaml")
env = EV2GymVec(config_file=cfg, num_envs=4096, state_function="V2G_profit_max_loads",
                reward_function="ProfitMax_TrPenalty_UserIncentives", seed=0, use_torch=False)
for s in (1, 2, 3):
    t0 = time.perf_counter(); env.reset(seed=s); env.engine.synchronize(); print(f"reset(seed={s}): {time.perf_counter() - t0:.2f} s")
t0 = time.perf_counter(); env.reset(); env.engine.synchronize(); print(f"reset(): {(time.perf_counter() - t0) * 1e3:.2f} ms")
