#!/usr/bin/env python3
"""Per-wavefront layer stamps of the float32 policy inside the fused launch (development tool; EV2G_LIB = a library built with -DEV2G_F32_STAMPS):
the LAST forward of one whole-episode ev2g_collect at cfg2, workgroups 0..7: cycles between the stamps, per wavefront."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ev2gym_amd import engine, _abi
from bench import WORKLOADS
from ev2gym_amd.actor import init_mlp_weights
from ev2gym_amd.scenario_gen import generate_native
WL = os.environ.get("STAMP_WORKLOAD", "cfg2")
wl = WORKLOADS[WL]
E = wl["envs"]
batch = generate_native(wl["gen"](E, 0)).sorted_by_busy_window(E)
eng = engine.Engine(batch, _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=_abi.FLAG_LOG_SOC)
P, D, T = eng.P, eng.D, eng.T
PREC = os.environ.get("STAMP_PRECISION", "fp32")   # (bf16: the same stamps in ev2g_mlp3_inline)
mlp = eng.mlp_create(*init_mlp_weights(D, P, seed=1), out_lo=wl["lo"], precision=PREC)
obs, act = eng.empty((T + 1, E, D), np.float32), eng.empty((T, E, P), np.float32)
rew, done, mask = eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
names = ["entry->barrier 1", "layer 1", "wait barrier 2", "layer 2", "wait barrier 3", "layer 3", "wait barrier 4"]
for rep in range(2):
    eng.reset_f32(obs, 0)
    eng.collect(mlp, T, obs, act, rew, done, mask)
    eng.synchronize()
    ms = eng.last_step_n_kernel_ms()
    out = (C.c_ulonglong * 3072)()
    eng._lib.ev2g_mlp_debug_f32_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    eng._lib.ev2g_mlp_debug_f32_stamps(eng._h, mlp, out)
    v2 = np.array(list(out)[1024:], np.int64).reshape(8, 16, 16)
    v = np.array(list(out)[:1024], np.int64).reshape(8, 16, 8)
print(f"fused {PREC} policy, {WL}, one launch of {T} steps, spec {eng.last_launch_specialisation}: {ms*1e3/T:.2f} us/step")
d = np.diff(v, axis=2)   # [wg][wave][7]
print("mean over 8 workgroups, cycles (s_memtime at 100 MHz? -> printed raw):")
print("wave " + " ".join(f"{n:>17s}" for n in names) + "   total")
for w in range(16):
    print(f"{w:4d} " + " ".join(f"{d[:, w, i].mean():17.0f}" for i in range(7)) + f"   {(v[:, w, 7] - v[:, w, 0]).mean():.0f}")
print("span (first entry -> last exit) per workgroup:", [int(v[g, :, 7].max() - v[g, :, 0].min()) for g in range(8)])
print("layer 3, wavefronts that own a tile: cycles from the layer's start (stamp 5) to the end of each k-step, then to the epilogue's start; workgroup 0..3")
for g in range(4):
    for w in range(16):
        if PREC != "bf16" and v2[g, w, 0] > v[g, w, 5]:
            print(f"  wg {g} wave {w:2d}: " + " ".join(f"{int(v2[g, w, i] - v[g, w, 5]):5d}" for i in list(range(10)) + [15]) + f"   layer end {int(v[g, w, 6] - v[g, w, 5])}")
