#!/usr/bin/env python3
"""How fast is the Python-level loop `obs, r, d, t, info = env.step(a)` itself (EV2GymVec, torch tensors, actions resident)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ev2gym_amd.vec_env import EV2GymVec
from ev2gym_amd.scenario_gen import GenConfig, generate
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = EV2GymVec(scenarios=generate(GenConfig.v2g_profit_plus_loads(E, 50, 1, seed=0)), state_function="V2G_profit_max_loads",
                reward_function="ProfitMax_TrPenalty_UserIncentives", auto_reset=True)
T, P = env.simulation_length, env.number_of_ports
acts = torch.rand((T, E, P), dtype=torch.float64, device="cuda") * 2 - 1
env.reset()
for t in range(T):
    env.step(acts[t])
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5 * T
for i in range(n):
    obs, rew, done, trunc, info = env.step(acts[i % T])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"EV2GymVec.step loop: {dt / n * 1e6:.1f} us/step, {E * n / dt / 1e6:.1f} M env-steps/s (E={E})")
