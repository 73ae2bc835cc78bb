#!/usr/bin/env python3
"""Per-phase cycle accounting of the fused actor + step launch (development tool): a private library built with -DEV2G_PHASE_TIMING,
one whole episode collected in one ev2g_collect call at cfg2; slot 7 = the policy (ev2g_mlp3_inline) including its first barrier.
  python tools/phase_timing_fused.py            (EV2G_PT_LIB=<prebuilt .so> to skip the build; EV2G_PT_BUILD_ONLY=1 to only build)"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ev2gym_amd import build, engine, _abi
so = os.path.abspath(os.environ.get("EV2G_PT_LIB") or os.path.join(ROOT, "build_variants", "pt_fused.so"))
if not os.environ.get("EV2G_PT_LIB"):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([build.hipcc()] + build.FLAGS + ["-DEV2G_PHASE_TIMING"] + [a for a in sys.argv[1:] if a.startswith("-D")] + ["-o", so, build.SRC])
if os.environ.get("EV2G_PT_BUILD_ONLY"):
    sys.exit(0)
engine._LIB_PATH = so
L = engine.load_library(so)
from bench import WORKLOADS
from ev2gym_amd.actor import init_mlp_weights
from ev2gym_amd.scenario_gen import generate_native
wl = WORKLOADS["cfg2"]
E = wl["envs"]
batch = generate_native(wl["gen"](E, 0)).sorted_by_busy_window(E)
eng = engine.Engine(batch, _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=_abi.FLAG_LOG_SOC)
P, D, T = eng.P, eng.D, eng.T
mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=1), out_lo=wl["lo"])
obs, act = eng.empty((T + 1, E, D), np.float32), eng.empty((T, E, P), np.float32)
rew, done, mask = eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
names = ["A home/charger", "barrier waits", "B battery maths", "C home", "D reduce", "E env-level", "event prefetch", "POLICY (3 layers, 4 barriers)"]
for rep in range(2):
    eng.reset_f32(obs, 0)
    eng.synchronize()
    L.ev2g_debug_phase_ticks(eng._h, (C.c_ulonglong * 18)())   # (reads and clears)
    eng.collect(mlp, T, obs, act, rew, done, mask)
    eng.synchronize()
    out = (C.c_ulonglong * 18)()
    L.ev2g_debug_phase_ticks(eng._h, out)
    v = np.array(list(out), float)
    ms = eng.last_step_n_kernel_ms()
    nb, ne = max(v[16], 1), max(v[17], 1)
    print(f"fused cfg2, one launch of {T} steps, spec {eng.last_launch_specialisation}: {ms*1e3/T:.2f} us/step; workgroup-steps with items {int(v[16])} ({v[:8].sum()/nb:.0f} ticks each), without {int(v[17])} ({v[8:16].sum()/ne:.0f} ticks each)")
    for i, n in enumerate(names):
        if v[i] or v[8 + i]:
            print(f"   {n:30s} busy {v[i]/nb:8.0f}   empty {v[8+i]/ne:8.0f}   ticks/workgroup-step")
