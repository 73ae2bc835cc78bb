#!/usr/bin/env python3
"""Per-phase cycles of the cfg2 persistent launch with the outputs overwritten in place (step stride 0) against every row kept (strided): where do the
+19 % go?  (development tool; EV2G_PT_LIB = a library built with -DEV2G_PHASE_TIMING [-DEV2G_PT_BSPLIT])"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ev2gym_amd import engine, _abi
so = os.path.abspath(os.environ["EV2G_PT_LIB"])
engine._LIB_PATH = so
L = engine.load_library(so)
from bench import WORKLOADS
from ev2gym_amd.scenario_gen import generate
wl = WORKLOADS["cfg2"]
E = wl["envs"]
eng = engine.Engine(generate(wl["gen"](E, 0)), _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=_abi.FLAG_LOG_SOC)
P, D, T = eng.P, eng.D, eng.T
acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
obs, rew, done, mask = eng.empty((T, E, D)), eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
names = ["A home/charger", "barrier waits", "B battery maths", "C home", "D reduce", "E env-level", "prefetch issue", "loop top"]
if os.environ.get("PT_BSPLIT"): names[7], names[2], names[0] = "B operands (LDS + record wait)", "B battery maths proper", "A home/charger + loop top"
for strided in (False, True, False, True):
    eng.reset(obs)
    s = (E * D, E, E, E * P) if strided else (0, 0, 0, 0)
    eng.step_n(T, acts, E * P, obs, s[0], rew, s[1], done, s[2], mask, s[3], auto_reset=False, persistent=True)
    eng.synchronize()
    out = (C.c_ulonglong * 18)()
    L.ev2g_debug_phase_ticks(eng._h, out)
    v = np.array(list(out), float)
    ms = eng.last_step_n_kernel_ms()
    nb, ne = max(v[16], 1), max(v[17], 1)
    print(f"cfg2 strided={strided} (specialisation {eng.last_launch_specialisation}): {ms*1e3/T:.2f} us/step; workgroup-steps with items {int(v[16])} ({v[:8].sum()/nb:.0f} ticks each), without {int(v[17])} ({v[8:16].sum()/ne:.0f} ticks each)")
    for i, n in enumerate(names):
        if v[i] or v[8 + i]:
            print(f"   {n:32s} busy {v[i]/nb:8.0f}   empty {v[8+i]/ne:8.0f}   ticks/workgroup-step")
