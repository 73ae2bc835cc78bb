import sys, numpy as np
sys.path.insert(0, '.')
from ev2gym_amd import _abi
from ev2gym_amd.engine import Engine
from ev2gym_amd.scenario_gen import GenConfig, generate_native
mk = lambda M, seed: GenConfig.v2g_profit_plus_loads(M, 50, 1, seed=seed)
b = generate_native(mk(24, 5))
rk, sk = 0, 0
for flags in (_abi.FLAG_LOG_SOC, _abi.FLAG_LOG_SOC | _abi.FLAG_REFILLABLE):
    print("load flags", flags, flush=True)
    eng = Engine(b, rk, sk, device=0, flags=flags)
    eng.synchronize(); print(" loaded, cap", eng.pool_session_capacity, flush=True)
    obs = eng.empty((eng.E, eng.D)); eng.reset(obs); eng.synchronize(); print(" reset ok", flush=True)
    st = eng.stats(); print(" stats ok", np.nan_to_num(st).sum(), flush=True)
    if flags & _abi.FLAG_REFILLABLE:
        eng.pool_refill(mk(24, 9), 9, 0, 0, 24); eng.synchronize(); print(" refill ok, overflows", eng.pool_refill_overflows, flush=True)
        eng.reset(obs); eng.synchronize(); print(" reset after refill ok", flush=True)
    eng.close()
