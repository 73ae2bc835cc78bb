#!/usr/bin/env python3
"""Throughput of the Stable-Baselines3 VecEnv protocol over the engine (numpy in / numpy out every step, as SB3 drives it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ev2gym_amd.sb3_vec_env import EV2GymSB3VecEnv
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ev2gym_amd", "example_config_files", "V2GProfitPlusLoads_50cs.yaml")
venv = EV2GymSB3VecEnv(config_file=cfg, num_envs=E, seed=0, state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives", copy_obs=os.environ.get("SB3_COPY_OBS", "1") != "0")
obs = venv.reset()
T, P = venv.vec.simulation_length, venv.vec.number_of_ports
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, (8, E, P)).astype(np.float32)
for t in range(T):
    venv.step(acts[t % 8])
n = 3 * T
t0 = time.perf_counter()
for i in range(n):
    obs, rew, done, infos = venv.step(acts[i % 8])
dt = time.perf_counter() - t0
print(f"EV2GymSB3VecEnv.step loop: {dt / n * 1e3:.3f} ms/step, {E * n / dt / 1e6:.2f} M env-steps/s (E={E}, obs {obs.dtype}{obs.shape}, infos {type(infos).__name__}[{len(infos)}])")
venv.close()
