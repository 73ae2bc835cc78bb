#!/usr/bin/env python3
"""Per-phase cycle accounting of the step kernels (development tool, not part of the product path).
Builds a private libev2g_hip_pt.so with -DEV2G_PHASE_TIMING [extra -D flags] and prints the cycles of each phase per
workgroup-step, separately for steps with and without battery-maths items.
  python tools/phase_timing.py [cfg2|cfg3|cfg4] [--outer] [-DMACRO ...]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ev2gym_amd import build, engine, _abi
so = os.path.join(ROOT, "gpurun_out", "libev2g_hip_pt.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
argv = sys.argv[1:]
OUTER = "--outer" in argv
defs = [a for a in argv if a.startswith("-D")]
argv = [a for a in argv if a != "--outer" and not a.startswith("-D")]
if os.environ.get("EV2G_PT_LIB"):   # a library built beforehand with the same flags (the build container cross-compiles; GPU minutes are for running)
    so = os.path.abspath(os.environ["EV2G_PT_LIB"])
else:
    subprocess.check_call([build.hipcc()] + build.FLAGS + ["-DEV2G_PHASE_TIMING"] + (["-DEV2G_PT_OUTER"] if OUTER else []) + defs + ["-o", so, build.SRC])
if os.environ.get("EV2G_PT_BUILD_ONLY"):
    sys.exit(0)
engine._LIB_PATH = so
L = engine.load_library(so)
from bench import WORKLOADS
from ev2gym_amd.scenario_gen import generate
wname = argv[0] if argv else "cfg2"
wl = WORKLOADS[wname]
E = int(os.environ.get("AB_ENVS", wl["envs"]))
batch = generate(wl["gen"](E, 0))
eng = engine.Engine(batch, _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=_abi.FLAG_LOG_SOC)
P, D, T = eng.P, eng.D, eng.T
acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
names = ["A home/charger", "barrier waits", "B battery maths", "C home", "D reduce", "E env-level", "prefetch issue", "loop top"]
if OUTER: names[6], names[7] = "EPILOGUE (state write-back)", "PROLOGUE (state load)"
if "-DEV2G_PT_BSPLIT" in defs: names[7], names[2], names[0] = "B operands (LDS + record wait)", "B battery maths proper", "A home/charger + loop top"
print(f"## {wname} {eng.kernel_name} defs={defs}")
for persistent in (True, False):
    eng.reset(obs)
    eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=persistent)
    eng.synchronize()
    out = (C.c_ulonglong * 18)()
    L.ev2g_debug_phase_ticks(eng._h, out)
    v = np.array(list(out), float)
    ms = eng.last_step_n_kernel_ms()
    nb, ne = max(v[16], 1), max(v[17], 1)
    print(f"{wname} persistent={persistent}: {ms*1e3/T:.2f} us/step; workgroup-steps with items {int(v[16])} ({v[:8].sum()/nb:.0f} ticks each), "
          f"without {int(v[17])} ({v[8:16].sum()/ne:.0f} ticks each)")
    for i, n in enumerate(names):
        if v[i] or v[8 + i]:
            print(f"   {n:28s} busy {v[i]/nb:8.0f}   empty {v[8+i]/ne:8.0f}   ticks/workgroup-step")
