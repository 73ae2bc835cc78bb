#!/usr/bin/env python3
"""A/B timing of builds of libev2g_hip.so (development tool): for every library given, in its own process, the step
kernel's duration (HIP events on the launch stream, median of N launches) for persistent 112-step launches and for
single-step launches, on the cfg2 / cfg3 workload with a resident scenario pool.

  python tools/ab_bench.py [--workload cfg2] [--pool 8] [--reps 30] lib1.so lib2.so ...        (parent)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(workload, pool, reps):
    import numpy as np
    from bench import WORKLOADS
    from ev2gym_amd import _abi
    from ev2gym_amd.engine import Engine
    from ev2gym_amd.scenario_gen import generate_native as generate, occupancy_fraction
    wl = WORKLOADS[workload]
    E = int(os.environ.get("AB_ENVS", wl["envs"]))
    M = E * pool
    gcfg = wl["gen"](M, 0)
    if os.environ.get("AB_SPAWN"):   # e.g. AB_SPAWN=0: no EV ever arrives, every step is a quiet step
        gcfg.spawn_multiplier = float(os.environ["AB_SPAWN"])
    batch = generate(gcfg)
    if os.environ.get("AB_SORT"):   # scenarios with similar busy windows next to each other (per window of E scenarios: every episode's env set)
        batch = batch.sorted_by_busy_window(E)
    phi = occupancy_fraction(batch)
    rk, sk = _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]]
    eng = Engine(batch, rk, sk, device=0, flags=_abi.FLAG_LOG_SOC, n_active_envs=E)
    P, D, T = eng.P, eng.D, eng.T
    acts = eng.empty((T, E, P)); eng.fill_uniform(acts, T * E * P, 1, wl["lo"], 1.0)
    obs, rew, done, mask = eng.empty((E, D)), eng.empty((E,)), eng.empty((E,), np.uint8), eng.empty((E, P), np.uint8)
    stats = eng.empty((E, _abi.N_STATS))
    io32 = bool(os.environ.get("AB_IO32"))   # the rollout's hand-over: float32 actions in, float32 observations out (no float64 ones)
    if io32:
        from ev2gym_amd.engine import host_uniform
        acts32 = eng.empty((T, E, P), np.float32).upload(host_uniform(T * E * P, 1, wl["lo"], 1.0).astype(np.float32))
        obs32 = eng.empty((E, D), np.float32)
        eng.set_extras(obs_f32=obs32, actions_f32=acts32)
    bes = P * (phi * wl["b_occ"] + (1 - phi) * wl["b_empty"]) + batch.n_transformers * wl["b_tr"] + wl["b_env"]
    out = {"lib": os.environ.get("AB_LABEL") or os.environ.get("EV2G_LIB", "default"), "kernel": eng.kernel_name}
    off = 0
    strided = bool(os.environ.get("AB_STRIDED"))   # persistent launches with every output kept: [T,E,*] blocks (instantiation 3)
    if strided:
        s_obs, s_rew, s_done, s_mask = eng.empty((T, E, D)), eng.empty((T, E)), eng.empty((T, E), np.uint8), eng.empty((T, E, P), np.uint8)
    for mode, persistent, n in (("persistent", True, reps), ("per_step", False, max(3, reps // 6))):
        ms = []
        for r in range(n + 2):
            off = (off + E) % M
            eng.reset(obs, offset=off)
            if strided and persistent:
                eng.step_n(T, acts, E * P, s_obs, E * D, s_rew, E, s_done, E, s_mask, E * P, auto_reset=False, persistent=True)
            else:
                eng.step_n(T, None if io32 else acts, E * P, None if io32 else obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=persistent)
            k = eng.last_step_n_kernel_ms()
            eng.stats(out=stats)
            if r >= 2:
                ms.append(k)
        med = float(np.median(ms))
        out[mode] = {"us_per_step": med * 1e3 / T, "frac": bes * E * T / (med / 1e3) / 1e9 / 8000.0, "min_us_per_step": float(np.min(ms)) * 1e3 / T}
    eng.check_faults()
    # a parity spot check against the checksum of the default library's run is done by the caller: here only a digest
    eng.reset(obs, offset=0)
    eng.step_n(T, None if io32 else acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
    st = eng.stats()
    out["digest"] = [float(np.nansum(st[:, i])) for i in (1, 2, 3, 12, 16)] + [float(obs.to_host().sum()), float(rew.to_host().sum())]
    print("AB " + json.dumps(out), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(a[1], int(a[2]), int(a[3]))
        sys.exit(0)
    wl, pool, reps, libs = "cfg2", 8, 30, []
    i = 0
    while i < len(a):
        if a[i] == "--workload": wl = a[i + 1]; i += 2
        elif a[i] == "--pool": pool = int(a[i + 1]); i += 2
        elif a[i] == "--reps": reps = int(a[i + 1]); i += 2
        else: libs.append(a[i]); i += 1
    rows = []
    for lib in libs or [""]:
        env = dict(os.environ)
        label = lib
        if "@" in lib:   # lib.so@VAR=value[,VAR=value]: the same library under other environment settings (e.g. EV2G_KERNEL=wave)
            lib, sets = lib.split("@", 1)
            for kv in sets.split(","):
                k_, v_ = kv.split("=", 1)
                env[k_] = v_
        env["AB_LABEL"] = os.path.basename(label)
        if lib:
            env["EV2G_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wl, str(pool), str(reps)], env=env, capture_output=True, text=True)
        for l in r.stderr.splitlines():
            if l.startswith("[ev2g]"):
                print(l)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        if not line:
            print(f"{lib}: FAILED\n{r.stdout[-800:]}\n{r.stderr[-1500:]}")
            continue
        d = json.loads(line[0][3:])
        rows.append(d)
        print(f"{d['lib'] or 'default':34s} {d['kernel']:22s} persistent {d['persistent']['us_per_step']:.3f} us/step ({d['persistent']['frac']:.4f})   "
              f"per_step {d['per_step']['us_per_step']:.3f} us ({d['per_step']['frac']:.4f})   digest {['%.10g' % x for x in d['digest']]}", flush=True)
    if rows:
        ref = rows[0]["digest"]
        for d in rows[1:]:
            same = all(abs(x - y) <= 1e-9 * max(1.0, abs(y)) for x, y in zip(d["digest"], ref))
            print(f"{os.path.basename(d['lib']):40s} digest {'==' if same else '!='} first library")
