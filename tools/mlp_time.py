#!/usr/bin/env python3
"""Stand-alone timing of the fused actor kernel (development tool): n back-to-back forwards, wall time per forward."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ev2gym_amd import _abi
from ev2gym_amd.actor import init_mlp_weights
from ev2gym_amd.engine import Engine
from ev2gym_amd.scenario_gen import GenConfig, generate
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pool = generate(GenConfig.v2g_profit_plus_loads(8, 50, 1, seed=1))
eng = Engine(pool, 0, 0, device=0)
for (D, P) in ((162, 50), (63, 20)):
    w = init_mlp_weights(D, P, seed=3)
    m = eng.mlp_create(*w, precision=os.environ.get("MLP_PREC", "bf16"))
    x = eng.empty((E, D), np.float32).upload(np.random.default_rng(0).normal(0, 1, (E, D)).astype(np.float32))
    y = eng.empty((E, P), np.float32)
    for n in (50, 400):
        eng.synchronize(); t0 = time.perf_counter()
        for _ in range(n): eng.mlp_forward(m, x, y, E)
        eng.synchronize(); dt = time.perf_counter() - t0
    print(f"mlp {D}->400->300->{P} [{os.environ.get('MLP_PREC', 'bf16')}], {E} rows: {dt / n * 1e6:.2f} us per forward (back to back, incl. launch)")
    eng.mlp_destroy(m)

