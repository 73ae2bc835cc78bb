import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from ev2gym_amd import _abi
from ev2gym_amd.engine import Engine
from bench import WORKLOADS
from ev2gym_amd.scenario_gen import generate
wl = WORKLOADS["cfg2"]; E = 4096
for flags, steps in ((_abi.FLAG_LOG_SOC, 112), (_abi.FLAG_LOG_SOC, 84), (_abi.FLAG_LOG_SOC, 56), (_abi.FLAG_LOG_SOC, 28), (0, 112)):
    eng = Engine(generate(wl["gen"](E, 0)), _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=flags)
    P, T = eng.P, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
    rew, done, mask, obs = eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8), eng.empty((E, eng.D))
    eng.reset(); eng.step_n(steps, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
    out = eng.empty((E, 17)); eng.stats(out); eng.synchronize()
    n = 40; t0 = time.perf_counter()
    for _ in range(n): eng.stats(out)
    eng.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"log_soc={int(flags != 0)} after {steps:3d} steps: statistics kernel {dt * 1e6:7.1f} us")
    eng.close()
