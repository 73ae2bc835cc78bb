#!/usr/bin/env python3
"""Cost of drawing scenarios on the device (development tool): ev2g_pool_refill of one window of the cfg2 / cfg3 pool, next to the host generator.
  python tools/refill_time.py [cfg2|cfg3]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS
from ev2gym_amd import _abi
from ev2gym_amd.engine import Engine
from ev2gym_amd.scenario_gen import generate_native
w = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = WORKLOADS[w]
E = wl["envs"]
cfg = wl["gen"](2 * E, 3)
pool = generate_native(cfg)   # (first call: loads the library)
t0 = time.perf_counter(); pool = generate_native(cfg); t_host = time.perf_counter() - t0   # ev2g_generate + the numpy copies / port resolution of the Python wrapper
eng = Engine(pool, _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], device=0, flags=_abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE, n_active_envs=E)
eng.pool_refill(cfg, 3, 2 * E, 0, E); eng.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    eng.pool_refill(cfg, 3, (3 + i) * E, (i & 1) * E, E)
eng.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{w}: device refill of {E} scenarios: {dt * 1e6:.1f} us = {E / dt / 1e6:.1f} M scenarios/s (session capacity {eng.pool_session_capacity}, overflows {eng.pool_refill_overflows}); "
      f"host generate_native (warm, default threads of {os.cpu_count()} cpus): {t_host / (2 * E) * 1e6:.2f} us per scenario = {2 * E / t_host / 1e6:.3f} M scenarios/s; tools/gen_host_scaling.py times ev2g_generate alone")
