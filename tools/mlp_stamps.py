#!/usr/bin/env python3
"""Per-phase cycle stamps of the fused actor kernel (workgroup 0, lane 0); development tool.  Run as
   EV2G_LIB=<lib built with -DEV2G_MLP_TIMING> python tools/mlp_stamps.py   (tools builds it first when EV2G_LIB is unset)"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ev2gym_amd import build
so = os.path.join(ROOT, "gpurun_out", "libev2g_hip_mlpt.so")
if not os.environ.get("EV2G_LIB"):   # (a library built elsewhere with -DEV2G_MLP_TIMING is taken as is)
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([build.hipcc()] + build.FLAGS + ["-DEV2G_MLP_TIMING", "-DEV2G_ONLY_00", "-o", so, build.SRC])
    os.environ["EV2G_LIB"] = so
    os.execv(sys.executable, [sys.executable] + sys.argv)
from ev2gym_amd.actor import init_mlp_weights
from ev2gym_amd.engine import Engine
from ev2gym_amd.scenario_gen import GenConfig, generate
eng = Engine(generate(GenConfig.v2g_profit_plus_loads(8, 50, 1, seed=1)), 0, 0, device=0)
E, D, P = int(os.environ.get("MLP_ROWS", "4096")), 162, 50
m = eng.mlp_create(*init_mlp_weights(D, P, seed=3))
x = eng.empty((E, D), np.float32).upload(np.random.default_rng(0).normal(0, 1, (E, D)).astype(np.float32))
y = eng.empty((E, P), np.float32)
names = ["input load+convert", "(sync)", "layer 1", "(sync)", "layer 2", "(sync)", "layer 3"]
for rep in range(3):
    eng.mlp_forward(m, x, y, E)
    out = (C.c_ulonglong * 16)()
    eng._lib.ev2g_mlp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    eng._lib.ev2g_mlp_debug_stamps(eng._h, m, out)
    v = list(out)
    print("run", rep, "total", v[7] - v[0], "cycles:", ", ".join(f"{n} {v[i+1]-v[i]}" for i, n in enumerate(names)))
    if v[8]:   # the 16-row kernel's finer prologue stamps: requests issued | weight head issued | input rows converted
        print("      prologue: to first weight request", v[8] - v[0], "| ring head issued", v[9] - v[8], "| input arrived + converted", v[10] - v[9], "| padding + biases", v[1] - v[10])
