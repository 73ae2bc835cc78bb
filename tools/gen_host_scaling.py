#!/usr/bin/env python3
"""Host generator (ev2g_generate) throughput against its thread count (development probe)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2gym_amd.scenario_gen import GenConfig, gen_config_c
from ev2gym_amd.engine import load_library
L = load_library()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cfg = GenConfig.v2g_profit_plus_loads(M, 50, 1, seed=3)
c, keep = gen_config_c(cfg)
print("cpus", os.cpu_count())
for n in (1, 1, 4, 16, 64, 0):
    res = C.c_void_p()
    t0 = time.perf_counter(); rc = L.ev2g_generate(C.byref(c), M, 3, n, C.byref(res)); dt = time.perf_counter() - t0
    print(f"threads {n:3d}: {dt:.3f} s for {M} scenarios = {dt / M * 1e6:.2f} us per scenario = {M / dt / 1e6:.3f} M scenarios/s", flush=True)
    L.ev2g_gen_free(res)
