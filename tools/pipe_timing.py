#!/usr/bin/env python3
"""Cycle accounting of ev2g_step_pipe (development tool): builds a private library with -DEV2G_PHASE_TIMING [-D...] and prints the
cycles per workgroup-step of each segment of the env wavefront 0 and of the first worker wavefront.
  python tools/pipe_timing.py [cfg2|cfg3] [-DMACRO ...]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ev2gym_amd import build, engine, _abi
so = os.path.join(ROOT, "gpurun_out", "libev2g_hip_pp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
argv = [a for a in sys.argv[1:] if not a.startswith("-D")]
subprocess.check_call([build.hipcc()] + build.FLAGS + ["-DEV2G_PHASE_TIMING", "-DEV2G_ONLY_00"] + defs + ["-o", so, build.SRC])
L = engine.load_library(so)
from bench import WORKLOADS
from ev2gym_amd.scenario_gen import generate
wname = argv[0] if argv else "cfg2"
wl = WORKLOADS[wname]
E = wl["envs"]
batch = generate(wl["gen"](E, 0))
eng = engine.Engine(batch, _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=_abi.FLAG_LOG_SOC)
P, D, T = eng.P, eng.D, eng.T
acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
names = ["top: vmcnt(0)", "grab + Cs", "A", "X wait", "P1 issue + Co", "D", "E", "P2 issue", "loop edge", "worker: X wait", "worker: B", "worker: Y wait"]
print(f"## {wname} {eng.launch_kernel_name(T, True)} defs={defs}")
for rep in range(2):
    eng.reset(obs)
    out = (C.c_ulonglong * 18)()
    L.ev2g_debug_phase_ticks(eng._h, out)
    eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
    eng.synchronize()
    L.ev2g_debug_phase_ticks(eng._h, out)
    ms = eng.last_step_n_kernel_ms()
v = np.array(list(out), float)
import math
nwg = math.ceil(E / (int(os.environ.get("PIPE_ENVW", "4")) * (64 // P)))
print(f"{ms*1e3/T:.2f} us/step; cycles per workgroup-step (sum over env segments {v[:9].sum()/nwg/T:.0f}, worker {v[9:12].sum()/nwg/T:.0f})")
for i, n in enumerate(names):
    print(f"   {n:18s} {v[i]/nwg/T:8.0f}")
