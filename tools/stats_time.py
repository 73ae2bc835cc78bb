#!/usr/bin/env python3
"""Time of the statistics kernel at the end of an episode (development probe; EV2G_LIB selects the library variant)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2gym_amd import _abi
from ev2gym_amd.engine import Engine
from bench import WORKLOADS
from ev2gym_amd.scenario_gen import generate
for wname in sys.argv[1:] or ["cfg2", "cfg3"]:
    wl = WORKLOADS[wname]; E = wl["envs"]
    eng = Engine(generate(wl["gen"](E, 0)), _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=_abi.FLAG_LOG_SOC)
    P, T = eng.P, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
    eng.reset(); eng.step_n(T, acts, E * P, None, 0, None, 0, None, 0, None, 0, auto_reset=False, persistent=True)
    out = eng.empty((E, 17)); eng.stats(out); eng.synchronize()
    ref = out.to_host().copy()
    n = 40; t0 = time.perf_counter()
    for _ in range(n): eng.stats(out)
    eng.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{os.environ.get('EV2G_LIB', 'default'):40s} {wname}: statistics kernel {dt * 1e6:7.1f} us   digest {np.nansum(ref):.12e}")
    eng.close()
