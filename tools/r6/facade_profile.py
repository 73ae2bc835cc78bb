#!/usr/bin/env python3
"""Where a step of the single-env facade spends its time (development tool)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ev2gym_amd.env import EV2Gym
from ev2gym_amd.baselines.heuristics import ChargeAsFastAsPossible
cfg = os.path.join(ROOT, "ev2gym_amd", "example_config_files", "V2GProfitPlusLoads.yaml")
env = EV2Gym(config_file=cfg, seed=1, state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives")
agent = ChargeAsFastAsPossible()
env.reset()
for t in range(50): env.step(agent.get_action(env))
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); n = 0
for t in range(50, env.simulation_length):
    env.step(agent.get_action(env)); n += 1
dt = time.perf_counter() - t0
pr.disable()
print(f"{dt / n * 1e3:.3f} ms/step")
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
