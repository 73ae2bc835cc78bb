#!/usr/bin/env python3
"""Where a VecEnv step of the SB3 adapter spends its time (development tool)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ev2gym_amd.sb3_vec_env import EV2GymSB3VecEnv
E = 4096
cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "ev2gym_amd", "example_config_files", "V2GProfitPlusLoads_50cs.yaml")
venv = EV2GymSB3VecEnv(config_file=cfg, num_envs=E, seed=0, state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives")
venv.reset()
P = venv.vec.number_of_ports
acts = np.random.default_rng(0).uniform(-1, 1, (8, E, P)).astype(np.float32)
for t in range(120): venv.step(acts[t % 8])
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(200): venv.step(acts[i % 8])
dt = time.perf_counter() - t0
pr.disable()
print(f"{dt / 200 * 1e3:.3f} ms/step = {E * 200 / dt / 1e6:.2f} M env-steps/s")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
