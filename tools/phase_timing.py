#!/usr/bin/env python3
"""Per-phase cycle accounting of ev2g_step_v2 (development tool, not part of the product path).
Builds a private libev2g_hip_pt.so with -DEV2G_PHASE_TIMING and prints the share of each phase."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ev2gym_amd import build, engine, _abi
so = os.path.join(ROOT, "gpurun_out", "libev2g_hip_pt.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
OUTER = "--outer" in sys.argv
if OUTER: sys.argv.remove("--outer")
subprocess.check_call([build.hipcc()] + build.FLAGS + ["-DEV2G_PHASE_TIMING"] + (["-DEV2G_PT_OUTER"] if OUTER else []) + ["-o", so, build.SRC])
engine._LIB_PATH = so
L = engine.load_library(so)
from bench import WORKLOADS
from ev2gym_amd.scenario_gen import generate
wname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = WORKLOADS[wname]
E = wl["envs"]
batch = generate(wl["gen"](E, 0))
eng = engine.Engine(batch, _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]])
P, D, T = eng.P, eng.D, eng.T
acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
names = ["A home/charger", "barrier waits", "B battery maths", "C home", "D reduce", "E env-level", "prefetch issue", "loop top"]
if OUTER: names[6], names[7] = "EPILOGUE (state write-back)", "PROLOGUE (state load)"
for persistent in (True, False):
    eng.reset(obs)
    eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=persistent)
    eng.synchronize()
    out = (C.c_ulonglong * 8)()
    L.ev2g_debug_phase_ticks(eng._h, out)
    v = np.array(list(out), float)
    ms = eng.last_step_n_kernel_ms()
    G = int(os.environ.get('EV2G_WB', '256')) // 64 * (64 // P) if (P <= 64 and batch.n_transformers == 1) else max(1, 256 // P)
    ng = (E + G - 1) // G
    tot = v.sum()
    print(f"{wname} persistent={persistent}: {ms*1e3/T:.2f} us/step, {tot/T/ng:.0f} ticks per workgroup-step")
    for n, x in zip(names, v):
        if x:
            print(f"   {n:28s} {100*x/tot:5.1f} %   {x/T/ng:9.0f} ticks/workgroup-step")
