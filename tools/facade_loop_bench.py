#!/usr/bin/env python3
"""Steps per second of the single-env facade (EV2Gym on a 1-env engine) in the reference's own usage patterns:
fused plugins with a numpy action array, and a heuristic that walks the object graph every step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ev2gym_amd.env import EV2Gym
from ev2gym_amd.baselines.heuristics import ChargeAsFastAsPossible
cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ev2gym_amd", "example_config_files", "V2GProfitPlusLoads.yaml")
for name, agent in (("np.ones actions", None), ("ChargeAsFastAsPossible (object graph)", ChargeAsFastAsPossible())):
    env = EV2Gym(config_file=cfg, seed=1, state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives")
    env.reset()
    n, t0 = 0, time.perf_counter()
    for ep in range(3):
        env.reset()
        for t in range(env.simulation_length):
            a = np.ones(env.number_of_ports) if agent is None else agent.get_action(env)
            env.step(a)
            n += 1
    dt = time.perf_counter() - t0
    print(f"EV2Gym facade, {name}: {n / dt:.0f} env-steps/s ({dt / n * 1e3:.2f} ms/step, 25 chargers; reference CPython: ~1075/s at 50 chargers)")
    env.close()
