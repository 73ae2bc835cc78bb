#!/usr/bin/env python3
"""bench.py -- env-steps/s of the vectorised EV2Gym step engine on N MI355X (one process per GPU).

A "step" is one batched EV2Gym.step(): every env of the rank's shard advances one timestep (4096 envs x
50 chargers per GPU at the default workload, BASELINE.json configs[1]).  Actions (uniform, RandomAgent
heuristics.py:546-558) are generated on the device BEFORE the timed region and stay resident in HBM; the
timed region contains the step kernels, the per-episode statistics kernel, the per-episode reset and -- for
N > 1 -- the RCCL all-gather of the episode statistics (the only collective on the path).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4] [--envs E_per_gpu]
                  [--launch per_step|persistent] [--actor mlp]

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ev2gym_amd import _abi  # noqa: E402
from ev2gym_amd.scenario_gen import GenConfig, generate_native, occupancy_fraction  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# name -> (generator config factory, reward, state, default envs/GPU, action low, bytes model)
# bytes model: per occupied / empty port-step, per (env,transformer)-step, per env-step
WORKLOADS = {
    "cfg2": dict(desc="V2GProfitPlusLoads, 50 chargers, 1 transformer, uniform[-1,1] actions",
                 gen=lambda E, seed: GenConfig.v2g_profit_plus_loads(E, 50, 1, seed=seed),
                 reward="ProfitMax_TrPenalty_UserIncentives", state="V2G_profit_max_loads", envs=4096, lo=-1.0,
                 b_occ=149, b_empty=33, b_tr=680, b_env=370),
    "cfg3": dict(desc="PublicPST, 20 chargers, SquaredTrackingErrorReward, uniform[0,1] actions",
                 gen=lambda E, seed: GenConfig.public_pst(E, 20, seed=seed),
                 reward="SquaredTrackingErrorReward", state="PublicPST", envs=8192, lo=0.0,
                 b_occ=157, b_empty=41, b_tr=40, b_env=60),
    "cfg4": dict(desc="synthetic 1000 chargers / 50 transformers, uniform[-1,1] actions",
                 gen=lambda E, seed: GenConfig.v2g_profit_plus_loads(E, 1000, 50, seed=seed),
                 reward="ProfitMax_TrPenalty_UserIncentives", state="V2G_profit_max_loads", envs=2048, lo=-1.0,
                 b_occ=149, b_empty=33, b_tr=680, b_env=370),
}


def measured_traffic(workload, launch, steps_per_launch, envs):
    """(bytes, source): HBM bytes per launch of the step kernel from the COMMITTED rocprofv3 PMC passes of the newest round that profiled this
    workload / launch shape (profiles/rNN_hbm_traffic.json, written by tools/prof_step.sh + tools/collect_evidence.py: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs, FETCH doubled per the gfx950 note in MI355X_MICROARCH.md) -- a stored figure of the same
    kernel on the same workload, NOT a measurement of this run (counters need rocprofv3 around the process): `roofline.traffic_source` names
    the file.  (None, None) when the shape was not profiled."""
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))[workload][launch]
        except Exception:
            continue
        if abs(t["steps_per_launch"] - steps_per_launch) > 1e-9 or envs != WORKLOADS[workload]["envs"]:
            continue   # (an older round may hold the matching shape)
        return (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0, "profiles/" + name + " (rocprofv3 PMC passes of an earlier run of this workload; not re-measured here)"
    return None, None


def host_cores():
    """Physical cores / sockets / logical CPUs of this host from /proc/cpuinfo (SURVEY par.8d asks for the core count next to the CPU figure)."""
    try:
        phys, sockets, logical = set(), set(), 0
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                logical += 1
            elif line.startswith("physical id"):
                pid = line.split(":")[1].strip(); sockets.add(pid)
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip(); phys.add((pid, cid))
        return dict(physical_cores=len(phys) or None, sockets=len(sockets) or None, logical_cpus=logical or os.cpu_count())
    except Exception:
        return dict(physical_cores=None, sockets=None, logical_cpus=os.cpu_count())


def cpu_baseline(batch, rk, sk, lo, budget_s=float(os.environ.get("EV2G_BENCH_CPU_BUDGET", "12.0"))):
    """The C oracle (a scalar port of the reference step(), oracle/ev2g_oracle.c) timed on this box's host
    cores, single thread, on whole episodes of a prefix of the same env batch."""
    from oracle.oracle import Oracle
    from ev2gym_amd.engine import host_uniform
    n = batch.n_envs
    sub = batch
    ora = Oracle(sub, rk, sk)
    E, P, T = sub.n_envs, sub.n_ports, sub.n_steps
    acts = host_uniform(T * E * P, 12345, lo, 1.0).reshape(T, E, P)
    obs = np.empty((E, ora.D))
    rew = np.empty(E)
    done = np.empty(E, np.uint8)
    mask = np.empty((E, P), np.uint8)
    steps = 0
    t_total = 0.0
    episodes = 0
    while t_total < budget_s and episodes < 64:
        ora.reset()
        a = acts.copy()
        t0 = time.perf_counter()
        for t in range(T):
            ora.step_range_nocopy(0, E, a[t], obs, rew, done, mask)
        t_total += time.perf_counter() - t0
        steps += E * T
        episodes += 1
    # same oracle, one thread per host core over disjoint env ranges (envs are independent; ctypes drops the GIL)
    import threading
    nthr = max(1, min(os.cpu_count() or 1, n // 8))
    bounds = np.linspace(0, E, nthr + 1).astype(int)
    mt_steps, mt_total, mt_eps = 0, 0.0, 0
    while nthr > 1 and mt_total < budget_s / 3 and mt_eps < 64:
        ora.reset()
        a = acts.copy()

        def work(i):   # one C call per thread and episode
            ora.run_range_nocopy(int(bounds[i]), int(bounds[i + 1]), T, a, E * P, obs, rew, done, mask)
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        mt_total += time.perf_counter() - t0
        mt_steps += E * T
        mt_eps += 1
    all_cores = dict(value=mt_steps / mt_total, cores=nthr, episodes=mt_eps, **host_cores(),
                     note="threads = min(logical CPUs, envs / 8); the 512-env sample gives each thread 8 envs per episode, so thread start-up and the "
                          "hyper-threads' shared cores keep the speed-up far below the thread count -- a bound on what the host reaches, not a tuned CPU port") if mt_eps else None
    ora.close()
    return dict(value=steps / t_total, unit="env-steps/s", cores=1, kind="port", all_cores=all_cores, host=host_cores(),
                sample=f"{n} envs x {episodes} episodes x {T} steps of the same workload, C oracle -O2, 1 thread; "
                       f"reference CPython step() measured at build time: 1075 env-steps/s/core at 50 chargers (BASELINE.md)")


class RolloutLoop:
    """The stepping loop the benchmark times (and tests/test_dist_gloo.py drives on CPU with a stub engine): whole
    episodes where possible; at every episode end the statistics kernel, the (asynchronous) gather of the statistics over
    the ranks and a reset onto the next window of the resident scenario pool."""

    def __init__(self, eng, E, P, T, M, acts, obs, rew, done, mask, stats, gath=None, actor=None):
        self.eng, self.E, self.P, self.T, self.M = eng, E, P, T, M
        self.acts, self.obs, self.rew, self.done, self.mask, self.stats = acts, obs, rew, done, mask, stats
        self.gath, self.actor = gath, actor
        self.offset = 0
        self.episodes = 0

    def reset(self):
        self.offset = (self.offset + self.E) % self.M   # fresh scenarios every episode (E <= M: no env shares one)
        self.eng.reset(self.obs, offset=self.offset)

    def episode_end(self):
        """Terminal statistics of the finished episode, then the reset onto the next pool window -- one kernel launch where the engine
        offers it (ev2g_get_stats_reset), two otherwise; the asynchronous gather over the ranks is ordered behind the statistics."""
        out = self.stats if self.gath is None else self.gath.buffer()
        fused = getattr(self.eng, "stats_reset", None)
        if fused is not None:
            self.offset = (self.offset + self.E) % self.M
            fused(out, self.obs, self.offset)
        else:
            self.eng.stats(out=out)
        if self.gath is not None:
            self.gath.launch()
        self.episodes += 1
        if fused is None:
            self.reset()

    def run(self, n_steps, persistent, timing=None):
        eng, T = self.eng, self.T
        left = n_steps
        while left > 0:
            t = eng.current_step
            k = min(left, T - t)
            if self.actor is not None:
                self.actor.run(self, k)   # k x (policy forward -> env step)
            else:
                eng.step_n(k, self.acts[t], self.E * self.P, self.obs, 0, self.rew, 0, self.done, 0, self.mask, 0,
                           auto_reset=False, persistent=persistent)
            if self.gath is not None and hasattr(self.gath, "flush"):
                self.gath.flush()   # the statistics gather of the episode before, submitted BEHIND this launch (dist.AsyncStatsGather, "deferred")
            if timing is not None:
                timing.append((eng.last_step_n_kernel_ms(), k))
            left -= k
            if eng.current_step >= T:
                self.episode_end()


def full_episode_pass(loop, T, barrier, agree_max, min_s=0.1):
    """Whole episodes (persistent launch + statistics + gather + reset) for at least `min_s`: returns (episodes, seconds per episode).
    Every episode end issues a collective (the statistics gather), so all ranks must run the SAME number of episodes: the count is
    fixed up front from one timed episode and agreed on (`agree_max`: MAX over the ranks) -- never decided by a rank's own clock
    inside the loop, which would leave the faster rank waiting in a collective the slower one never issues."""
    loop.reset()
    loop.run(T, True)
    barrier()
    t0 = time.perf_counter()
    if loop.eng.current_step:
        loop.reset()
    loop.run(T, True)
    barrier()
    n_ep = min(agree_max(max(3, int(np.ceil(min_s / max(time.perf_counter() - t0, 1e-6))))), 2000)
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_ep):
        if loop.eng.current_step:
            loop.reset()
        loop.run(T, True)
    barrier()
    return n_ep, (time.perf_counter() - t0) / n_ep


class PipelinedLoops:
    """Several RolloutLoops over disjoint env groups of one GPU, each with its own engine handle and HIP stream, fed by one host
    thread per group.  With a policy in the loop every step is a chain of two dependent kernels (actor forward -> env step); two
    groups let the GPU run one group's actor forward next to the other group's env step, and hide the launch gaps of both."""

    def __init__(self, loops):
        from concurrent.futures import ThreadPoolExecutor
        self.loops = loops
        self.eng, self.T = loops[0].eng, loops[0].T
        self.pool = ThreadPoolExecutor(len(loops))

    def reset(self):
        for l in self.loops:
            l.reset()

    @property
    def episodes(self):
        return self.loops[0].episodes

    def run(self, n_steps, persistent, timing=None):
        futs = [self.pool.submit(l.run, n_steps, persistent, timing if i == 0 else None) for i, l in enumerate(self.loops)]
        for f in futs:
            f.result()


def rollout_record(Engine, batch, rk, sk, local_rank, devx, E, lo, rank, args, T, bytes_env_step, min_s=0.2, precision="bf16"):
    """BASELINE configs[4] shape per GPU: the fused actor (obs -> 400 -> 300 -> P, tanh) produces the actions on the device
    between single-step env launches (`ev2g_rollout`).  Returns env-steps/s of this rank and both kernel durations."""
    import torch
    from ev2gym_amd.actor import FusedMLPActor
    st = devx.new_stream(make_current=False)
    eng = Engine(batch, rk, sk, device=local_rank, stream=st, flags=0 if args.no_soc_log else _abi.FLAG_LOG_SOC, n_active_envs=E)
    try:
        dev, P, D = devx.device, eng.P, eng.D
        obs = torch.empty((E, D), dtype=torch.float64, device=dev)
        rew = torch.empty((E,), dtype=torch.float64, device=dev)
        done = torch.empty((E,), dtype=torch.uint8, device=dev)
        mask = torch.empty((E, P), dtype=torch.uint8, device=dev)
        stats = torch.empty((E, _abi.N_STATS), dtype=torch.float64, device=dev)
        actor = FusedMLPActor(eng, E, P, D, lo, dev, seed=1234 + rank, precision=precision)
        loop = RolloutLoop(eng, E, P, T, eng.M, None, obs, rew, done, mask, stats, None, actor)
        loop.reset()
        loop.run(T, False)
        eng.synchronize()
        n, spent = 0, 0.0
        while spent < min_s:
            t0 = time.perf_counter()
            loop.run(2 * T, False)
            eng.synchronize()
            spent += time.perf_counter() - t0
            n += 2 * T
        loop.reset()   # one whole episode as ONE rollout segment: the kernel time per step of whatever ev2g_rollout launches (round 5: one fused launch)
        eng.rollout(actor.mlp, T, rew, 0, done, 0, mask, 0)
        seg_us, seg_spec = eng.last_step_n_kernel_ms() * 1e3 / T, eng.last_launch_specialisation
        loop.reset()   # HIP events around a train of 64 single-step launches fed by the actor's action buffer (one event pair per train)
        eng.step_n(64, None, 0, None, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=False)
        step_us = eng.last_step_n_kernel_ms() * 1e3 / 64
        actor_us = actor.forward_train_us(200)
        eng.check_faults()
        return {"env_steps_per_s_per_gpu": E * n / spent, "us_per_step_wall": spent / n * 1e6, "precision": precision,
                "fused_launch": seg_spec == 4, "launches_per_segment": 1 if seg_spec == 4 else "2 per step", "segment_kernel_us_per_step": seg_us,
                "segment_roofline_frac_env_bytes_only": bytes_env_step * E / (seg_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                "step_kernel_us": step_us, "actor_kernel_us": actor_us, "step_kernel": eng.kernel_name,
                "step_kernel_roofline_frac": bytes_env_step * E / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                "actor": actor.describe, "steps_timed": n,
                "note": "the policy between steps (the RL-loop mode).  Round 5: ev2g_rollout issues ONE launch per segment -- the policy runs inside the "
                        "step kernel's launch (ev2g_step_wave<.., 1024, true>) -- where the shape is eligible (`fused_launch`); `segment_kernel_us_per_step` is that "
                        "launch by HIP events.  step_kernel_us / actor_kernel_us: the two kernels of the unfused chain (EV2G_NO_FUSED=1, foreign policies), HIP "
                        "events around single-step env launches / a back-to-back train of 200 actor forwards"}
    finally:
        eng.close()


def collector_record(Engine, batch, rk, sk, local_rank, devx, E, lo, rank, args, T, min_s=0.2):
    """The off-policy collection loop of an SB3 DDPG run with the replay buffer on the device (sb3_vec_env.DeviceReplayCollector over ev2g_collect):
    actor forward -> env step with every transition written in place, statistics + reset at the episode ends; no host copy of observations."""
    from ev2gym_amd.actor import init_mlp_weights
    from ev2gym_amd.sb3_vec_env import DeviceReplayCollector
    st = devx.new_stream(make_current=False)
    eng = Engine(batch, rk, sk, device=local_rank, stream=st, flags=0 if args.no_soc_log else _abi.FLAG_LOG_SOC, n_active_envs=E)
    try:
        col = DeviceReplayCollector(eng, init_mlp_weights(eng.D, eng.P, seed=1234 + rank), lo, capacity_episodes=2, use_torch=False)
        col.collect_episode()
        eng.synchronize()
        n, spent = 0, 0.0
        while spent < min_s:
            t0 = time.perf_counter()
            col.collect_episode()
            eng.synchronize()
            spent += time.perf_counter() - t0
            n += 1
        eng.check_faults()
        spec = eng.last_launch_specialisation
        col.close()
        return {"env_steps_per_s_per_gpu": E * T * n / spent, "us_per_step_wall": spent / (n * T) * 1e6, "episodes_timed": n, "step_kernel_specialisation": spec,
                "replay_block_mb": round(((T + 1) * E * eng.D * 4 + T * E * (eng.P * 5 + 9)) / 1e6, 1),
                "note": "DeviceReplayCollector: k x (fused actor forward -> single-step env launch) with observation / action / reward / done / mask rows "
                        "written in place into a device-resident episode block (next_obs[t] = obs[t + 1]); get_statistics + reset in one launch at every episode end"}
    finally:
        eng.close()


def refill_record(Engine, wl, rk, sk, local_rank, devx, E, rank, args, T, min_s=0.1):
    """Scenario generation ON THE DEVICE (ev2g_pool_refill: EV2Gym.reset()'s per-episode draw, ev2gym_env.py:243-296, without host
    work): a pool of 3 windows drawn by the library's generator; every episode steps one window while the window of the episode
    before is re-drawn in place.  Returns the cost of re-drawing one window and the episode rate with perpetual fresh scenarios."""
    import torch
    from ev2gym_amd.scenario_gen import generate_native
    cfg = wl["gen"](3 * E, 7000 + rank)
    st = devx.new_stream(make_current=False)
    eng = Engine(generate_native(cfg), rk, sk, device=local_rank, stream=st, n_active_envs=E,
                 flags=(0 if args.no_soc_log else _abi.FLAG_LOG_SOC) | _abi.FLAG_REFILLABLE)
    try:
        dev, P, D = devx.device, eng.P, eng.D
        acts = torch.empty((T, E, P), dtype=torch.float64, device=dev)
        eng.fill_uniform(acts, T * E * P, 77 + rank, wl["lo"], 1.0)
        obs = torch.empty((E, D), dtype=torch.float64, device=dev)
        rew = torch.empty((E,), dtype=torch.float64, device=dev)
        done = torch.empty((E,), dtype=torch.uint8, device=dev)
        mask = torch.empty((E, P), dtype=torch.uint8, device=dev)
        stats = torch.empty((E, _abi.N_STATS), dtype=torch.float64, device=dev)
        nxt = 3 * E
        eng.pool_refill(cfg, cfg.seed, nxt, 0, E); nxt += E   # (first call: uploads the config's tables)
        eng.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            eng.pool_refill(cfg, cfg.seed, nxt, (i % 3) * E, E); nxt += E
        eng.synchronize()
        refill_us = (time.perf_counter() - t0) / 10 * 1e6

        def episode(k, refill):
            # episode k steps window k % 3; its statistics and the reset onto window (k + 1) % 3 are ONE launch (ev2g_get_stats_reset, like the timed
            # region's full_episode); the window of the episode before, (k + 2) % 3, gets new scenarios in stream order behind them
            eng.step_n(T, acts, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=True)
            eng.stats_reset(stats, obs, offset=((k + 1) % 3) * E)
            if refill:
                nonlocal_nxt[0] += E
                eng.pool_refill(cfg, cfg.seed, nonlocal_nxt[0], ((k + 2) % 3) * E, E)
        nonlocal_nxt = [nxt]
        rates = {}
        for refill in (False, True):
            eng.reset(obs, offset=0)
            episode(0, refill); eng.synchronize()
            n, spent = 0, 0.0
            while spent < min_s:
                t0 = time.perf_counter()
                for k in range(6):
                    episode(n + k, refill)
                eng.synchronize()
                spent += time.perf_counter() - t0
                n += 6
            rates[refill] = spent / n
        eng.check_faults()
        return {"us_per_window": refill_us, "scenarios_per_window": E, "scenarios_per_s": E / (refill_us * 1e-6),
                "ms_per_episode_without_refill": rates[False] * 1e3, "ms_per_episode_with_refill": rates[True] * 1e3,
                "env_steps_per_s_with_refill_per_gpu": E * T / rates[True], "truncated_scenarios": eng.pool_refill_overflows,
                "session_slots_per_scenario": eng.pool_session_capacity,
                "note": "every episode runs scenarios never stepped before, drawn on the device (bit-identical to ev2g_generate); "
                        "whole episodes = 112-step persistent launch + statistics and reset onto the next window in one launch (+ ev2g_pool_refill of one window)"}
    finally:
        eng.close()


def workload_bytes_env_step(wl, P, R, phi):
    return P * (phi * wl["b_occ"] + (1 - phi) * wl["b_empty"]) + R * wl["b_tr"] + wl["b_env"]


def kernel_roofline(bytes_env_step, E, steps_per_launch, launch_s, kernel):
    achieved = bytes_env_step * E * steps_per_launch / launch_s / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "kernel": kernel,
            "avg_launch_us": launch_s * 1e6, "steps_per_launch": steps_per_launch, "algorithmic_bytes_per_env_step": bytes_env_step}


def launched_kernel(eng):
    """Name of the kernel the last launch ran: big envs run their specialised launches on ev2g_step_big (specialisation 5) although the handle's
    general kernel is ev2g_step_v2<1024>."""
    return "ev2g_step_big<512>" if getattr(eng, "last_launch_specialisation", -1) == 5 else eng.kernel_name


def other_workload_record(Engine, name, E, local_rank, devx, rank, args, min_s=0.12):
    """One of the OTHER BASELINE single-GPU configs on the default line (configs[2] / configs[3]): its own scenario pool (two windows), whole
    episodes -- persistent 112-step launch + statistics + reset onto the other window -- for `min_s` seconds; the step kernel's launch
    duration by HIP events on the launch stream (ev2g_last_step_n_kernel_ms), the rate by the wall clock around synchronised episodes."""
    import torch
    wl = WORKLOADS[name]
    rk, sk = _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]]
    batch = generate_native(wl["gen"](2 * E, 4000 + rank)).sorted_by_busy_window(E)
    phi = occupancy_fraction(batch)
    eng = Engine(batch, rk, sk, device=local_rank, stream=devx.new_stream(make_current=False), n_active_envs=E,
                 flags=0 if args.no_soc_log else _abi.FLAG_LOG_SOC)
    try:
        dev, P, D, T = devx.device, eng.P, eng.D, eng.T
        acts = torch.empty((T, E, P), dtype=torch.float64, device=dev)
        eng.fill_uniform(acts, T * E * P, 555 + rank, wl["lo"], 1.0)
        obs = torch.empty((E, D), dtype=torch.float64, device=dev)
        rew = torch.empty((E,), dtype=torch.float64, device=dev)
        done = torch.empty((E,), dtype=torch.uint8, device=dev)
        mask = torch.empty((E, P), dtype=torch.uint8, device=dev)
        stats = torch.empty((E, _abi.N_STATS), dtype=torch.float64, device=dev)
        loop = RolloutLoop(eng, E, P, T, eng.M, acts, obs, rew, done, mask, stats)
        loop.reset(); loop.run(T, True); eng.synchronize()
        n, spent, B = 0, 0.0, 8
        while spent < min_s or n < 2 * B:   # episodes queued eight at a time (no host synchronisation inside a batch, like the headline's timed region)
            t0 = time.perf_counter()
            loop.run(B * T, True)
            eng.synchronize()
            spent += time.perf_counter() - t0
            n += B
        eng.check_faults()
        if hasattr(eng, "step_n_kernel_ms_back"):
            launch_s = float(np.mean([eng.step_n_kernel_ms_back(i) for i in range(B)])) / 1e3   # the last batch's eight launches, by their own HIP events
        else:
            tim = []
            loop.run(T, True, timing=tim)
            launch_s = tim[0][0] / 1e3
        bes = workload_bytes_env_step(wl, P, batch.n_transformers, phi)
        extra = {}
        if name == "cfg3" and not args.no_rollout_record:   # round 6: the fused actor + step launch covers PublicPST (two envs per wavefront): its device-resident collector on the driver's line
            try:
                extra["collector"] = collector_record(Engine, batch, rk, sk, local_rank, devx, E, wl["lo"], rank, args, T, min_s=0.15)
                extra["collector"]["actor"] = f"fused MLP {D}->400->300->{P} tanh, bf16 operands on the matrix cores, inside the step kernel's launch when step_kernel_specialisation == 4"
            except Exception as ex:   # (the stand-in engines of the CPU tests have no collector)
                extra["collector"] = {"error": str(ex)}
        return {**extra, "workload": f"{name}: {wl['desc']}", "envs_per_gpu": E, "chargers": batch.n_chargers, "transformers": batch.n_transformers, "obs_dim": D,
                "occupancy_phi": round(phi, 4), "value": E * T * n / spent, "unit": "env-steps/s", "ms_per_step": spent / (n * T) * 1e3,
                "ms_per_episode": spent / n * 1e3, "episodes_timed": n, "launch": "persistent", "specialisation": eng.last_launch_specialisation,
                "roofline": kernel_roofline(bes, E, T, launch_s, launched_kernel(eng)),
                "contains": "whole episodes: 112-step persistent launch + statistics kernel + reset onto fresh scenarios (wall clock around batches of eight queued episodes); "
                            "roofline from the step kernel's own HIP-event duration (mean over the last batch's launches)"}
    finally:
        eng.close()


def strided_record(eng, loop, E, P, D, T, dev, bytes_env_step, stride0_launch_us, reps=5):
    """The persistent launch whose outputs are all KEPT: observation / reward / done / mask blocks [T, E, *] with step strides (a trajectory
    recorder's or a replay block's shape, generate_trajectories.py:69-83) instead of the headline's stride-0 rows that each step overwrites."""
    import torch
    obs = torch.empty((T, E, D), dtype=torch.float64, device=dev)
    rew = torch.empty((T, E), dtype=torch.float64, device=dev)
    done = torch.empty((T, E), dtype=torch.uint8, device=dev)
    mask = torch.empty((T, E, P), dtype=torch.uint8, device=dev)
    ms = []
    for r in range(reps + 1):
        loop.reset()
        eng.step_n(T, loop.acts, E * P, obs, E * D, rew, E, done, E, mask, E * P, auto_reset=False, persistent=True)
        k = eng.last_step_n_kernel_ms()
        eng.stats(out=loop.stats)
        if r:
            ms.append(k)
    spec = eng.last_launch_specialisation
    loop.reset()
    launch_s = float(np.median(ms)) / 1e3
    out = {"outputs": f"obs [T,E,{D}] f64 + reward [T,E] + done [T,E] + mask [T,E,{P}], every step kept ({obs.numel() * 8 / 1e6:.0f} MB of observations per launch)",
           "specialisation": spec, "us_per_step": launch_s * 1e6 / T, "env_steps_per_s_kernel_only": E * T / launch_s,
           "roofline": kernel_roofline(bytes_env_step, E, T, launch_s, eng.kernel_name),
           "vs_stride0_kernel_time": (launch_s * 1e6 / stride0_launch_us) if stride0_launch_us else None}
    del obs, mask
    return out


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` with no rendezvous in the environment: re-run this file as N ranks (one per GPU) under
    torch.distributed.run on this node -- the command the driver would type itself -- and pass rank 0's JSON line through."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


class Device:
    """The few torch.cuda calls the benchmark makes, behind one switch: `cuda` (the product: HIP streams, RCCL) or `cpu`
    (`--backend gloo`: tests/test_bench_launcher.py drives the launcher and the whole multi-rank control flow of this file
    with a stand-in engine on a box without a GPU; its line is marked as such and is never a measurement)."""

    def __init__(self, backend, local_rank):
        import torch
        self.torch, self.cuda, self.local_rank = torch, backend == "nccl", local_rank
        self.device = torch.device("cuda", local_rank) if self.cuda else torch.device("cpu")
        if self.cuda:
            torch.cuda.set_device(local_rank)

    def new_stream(self, make_current):
        if not self.cuda:
            return None
        st = self.torch.cuda.Stream(device=self.local_rank)
        if make_current:
            self.torch.cuda.set_stream(st)
        self._keep = getattr(self, "_keep", []) + [st]
        return st.cuda_stream

    def synchronize(self):
        if self.cuda:
            self.torch.cuda.synchronize()


def load_engine_class(spec):
    if not spec:
        from ev2gym_amd.engine import Engine
        return Engine
    import importlib
    mod, name = spec.split(":")
    return getattr(importlib.import_module(mod), name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1120)
    ap.add_argument("--warmup", type=int, default=112)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the workload's)")
    ap.add_argument("--keep-pool-order", action="store_true", help="do not co-schedule scenarios with similar busy windows (ScenarioBatch.sorted_by_busy_window)")
    ap.add_argument("--pool", type=int, default=0, help="scenario pool size as a multiple of --envs (default 8; cfg4: 2): every "
                    "episode steps a fresh window of the resident pool (the per-reset scenario draw of the reference)")
    ap.add_argument("--launch", default="auto", choices=["auto", "per_step", "persistent"])
    ap.add_argument("--min-time", type=float, default=0.25, help="repeat the --steps-sized timed region until this many seconds "
                    "have been measured per launch mode (median over the repetitions is reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-soc-log", action="store_true", help="skip the SoC log that the battery-degradation statistics need")
    ap.add_argument("--actor-groups", type=int, default=1, help="with --actor mlp: split the envs of a GPU into this many independently "
                    "pipelined groups (own engine handle + HIP stream each); 1 = one chain of dependent kernels.  Measured on MI355X: "
                    "1 -> 147 M, 2 -> 148-149 M, 4 -> 143-145 M env-steps/s (the gaps between dependent kernels are GPU-side)")
    ap.add_argument("--actor", default="none", choices=["none", "mlp", "mlp_fp32", "mlp_torch"],
                    help="BASELINE configs[4]-shaped rollout: an actor (obs->400->300->P, tanh; SB3-DDPG shape, random weights) "
                         "produces the actions on the device between steps (forces per_step launches).  mlp: the fused one-kernel "
                         "forward of the library (bf16 MFMA); mlp_fp32: the same kernel with float32 weights held as two bf16 terms (what an "
                         "SB3-trained float32 policy computes, to 1e-5); mlp_torch: the same network through torch.nn (fp32)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-rollout-record", action="store_true", help="skip the short policy-in-the-loop pass (`rollout` in the line)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short cfg3 / cfg4 passes (`other_workloads` in the cfg2 line)")
    ap.add_argument("--only-timed", action="store_true", help="profiling runs: nothing but the timed regions of the chosen launch mode "
                    "(no whole-episode pass, no HIP-event roofline pass, no rollout record), so that a kernel trace of `--launch per_step` "
                    "holds single-step dispatches only")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)   # gloo: CPU launcher test only
    ap.add_argument("--engine", default="", help=argparse.SUPPRESS)   # module:Class of a stand-in engine (tests only)
    args = ap.parse_args()
    if args.actor != "none":
        args.launch = "per_step"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries ONE line, the JSON: libraries that print banners on it (RCCL does at communicator creation) are sent to stderr
    # for the duration of the run -- file descriptor 1 itself, so that C-level printf is covered too
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the rendezvous in the environment has WORLD_SIZE={world}")
    stub = bool(args.engine) or args.backend != "nccl"
    devx = Device(args.backend, local_rank)
    # EV2G_BENCH_FORCE_DIST=1: run the multi-process code path (process group, asynchronous gather, C-ABI gather) at any world
    # size -- under torch.distributed.run with ONE process it exercises, on a single GPU, exactly what the N-GPU launch runs
    multi = world > 1 or bool(os.environ.get("EV2G_BENCH_FORCE_DIST"))
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if devx.cuda:
            dist.init_process_group("nccl", device_id=devx.device)
        else:
            dist.init_process_group("gloo")

    Engine = load_engine_class(args.engine)
    wl = WORKLOADS[args.workload]
    E = args.envs or wl["envs"]            # per GPU: weak scaling
    pool = args.pool or (2 if args.workload == "cfg4" else 8)
    M = E * pool
    rk, sk = _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]]
    # every rank draws its own pool of scenarios, with the library's own generator (ev2g_generate: the stream ev2g_pool_refill continues on the device)
    batch = generate_native(wl["gen"](M, args.seed * 1000 + rank))
    if not args.keep_pool_order:
        # The order of an i.i.d. pool is arbitrary; inside every window of E scenarios (one episode's env set) scenarios with similar busy
        # windows sit next to each other, so that the few envs a workgroup advances in lockstep wake up and fall idle together (a workgroup
        # without work skips its battery-maths phase): cfg2 -1.5 %, cfg3 -4 % kernel time (profiles/r04_ab_sorted_pool.txt)
        batch = batch.sorted_by_busy_window(E)
    phi = occupancy_fraction(batch)
    # engine kernels, torch allocations and the RCCL gather all run on ONE explicit (non-default) stream
    dev = devx.device
    n_groups = args.actor_groups if (args.actor in ("mlp", "mlp_fp32") and args.actor_groups > 1 and E % args.actor_groups == 0) else 1
    Eg, Mg = E // n_groups, M // n_groups
    gath = None
    if multi:   # double-buffered asynchronous all-gather: the statistics travel while the next episode steps
        from ev2gym_amd.dist import AsyncStatsGather
        gath = AsyncStatsGather(E, world, dev)
    loops, engines, actor = [], [], None
    for gi in range(n_groups):
        # engine kernels, torch allocations and the RCCL gather of a group all run on ONE explicit (non-default) stream
        gb = batch if n_groups == 1 else batch.select(np.arange(gi * Mg, (gi + 1) * Mg))
        eng = Engine(gb, rk, sk, device=local_rank, stream=devx.new_stream(make_current=(gi == 0)),
                     flags=0 if args.no_soc_log else _abi.FLAG_LOG_SOC, n_active_envs=Eg)
        P, D, T = eng.P, eng.D, eng.T
        acts = None
        if args.actor == "none":
            acts = torch.empty((T, Eg, P), dtype=torch.float64, device=dev)
            eng.fill_uniform(acts, T * Eg * P, 999 + rank, wl["lo"], 1.0)
        obs = torch.empty((Eg, D), dtype=torch.float64, device=dev)
        rew = torch.empty((Eg,), dtype=torch.float64, device=dev)
        done = torch.empty((Eg,), dtype=torch.uint8, device=dev)
        mask = torch.empty((Eg, P), dtype=torch.uint8, device=dev)
        stats = torch.empty((Eg, _abi.N_STATS), dtype=torch.float64, device=dev)
        g_actor = None
        if args.actor != "none":
            from ev2gym_amd.actor import make_actor
            g_actor = make_actor(eng, Eg, P, D, wl["lo"], dev, seed=1234 + rank, kind={"mlp": "fused", "mlp_fp32": "fused_fp32"}.get(args.actor, "torch"))
            actor = actor or g_actor
        loops.append(RolloutLoop(eng, Eg, P, T, Mg, acts, obs, rew, done, mask, stats, gath if n_groups == 1 else None, g_actor))
        engines.append(eng)
    eng = engines[0]
    loop = loops[0] if n_groups == 1 else PipelinedLoops(loops)

    def barrier():
        if gath is not None:
            gath.finish()
        devx.synchronize()
        if multi:
            dist.barrier()
        devx.synchronize()

    def timed(persistent):
        """Median over repetitions of the --steps-sized region.  One repetition = a chain of whole --steps regions long
        enough to out-weigh the launch / synchronisation edges (>= 20 ms), bracketed by barrier + synchronize; episode ends
        (statistics kernel, gather, reset onto fresh scenarios) fall where they fall inside the chain."""
        loop.reset()
        loop.run(args.warmup, persistent)
        loop.reset()
        barrier()
        t0 = time.perf_counter()
        loop.run(args.steps, persistent)
        barrier()
        one = time.perf_counter() - t0
        inner = max(1, int(np.ceil(0.02 / max(one, 1e-6))))
        if multi:   # every rank must run the same number of steps
            tt = torch.tensor([inner], dtype=torch.int64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            inner = int(tt.item())
        samples, spent = [], 0.0
        while spent < args.min_time or len(samples) < 3:
            barrier()
            t0 = time.perf_counter()
            loop.run(args.steps * inner, persistent)
            barrier()
            dt_ = time.perf_counter() - t0
            if multi:
                tt = torch.tensor([dt_], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_ = float(tt.item())
            samples.append(dt_ / inner)
            spent += dt_
            if len(samples) >= 200:
                break
        return float(np.median(samples)), len(samples), inner, samples

    modes = ["per_step", "persistent"] if args.launch == "auto" else [args.launch]
    res = {m: timed(m == "persistent") for m in modes}
    wall = {m: r[0] for m, r in res.items()}
    best = min(wall, key=wall.get)

    # whole episodes, the RL-free upper bound of the path: one 112-step persistent launch + statistics + reset per episode
    full_ep = None
    if actor is None and not args.only_timed:
        def agree_max(n):
            if not multi:
                return n
            tt = torch.tensor([n], dtype=torch.int64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return int(tt.item())
        n_ep, ep_s = full_episode_pass(loop, T, barrier, agree_max)
        full_ep = {"episodes": n_ep, "ms_per_episode": ep_s * 1e3, "env_steps_per_s": world * E * T / ep_s,
                   "contains": "112-step persistent launch + statistics kernel + reset onto fresh scenarios"}

    C_, R_ = batch.n_chargers, batch.n_transformers
    bytes_env_step = workload_bytes_env_step(wl, P, R_, phi)

    def roofline(mode):
        # kernel-only duration, HIP events on the launch stream around every ev2g_step_n of an untimed pass
        loop.reset()
        tim = []
        # (persistent: 16 whole-episode launches -- one or two, as in rounds 1-4, sample the first launches after a reset, 1 % slower than the
        # average rocprofv3 reports over the timed region's hundreds; per_step: two episodes = 224 launches)
        if mode == "persistent" and hasattr(eng, "step_n_kernel_ms_back") and n_groups == 1:
            # queued back to back like the timed region (no host synchronisation between the episodes); the handle keeps the HIP-event pairs of
            # its last 32 launches: read afterwards
            loop.run(16 * T, True)
            devx.synchronize()
            tim = [(eng.step_n_kernel_ms_back(i), T) for i in range(16)]
        else:
            loop.run(16 * T if mode == "persistent" else 2 * T, mode == "persistent", timing=tim)
        devx.synchronize()
        kern_ms = sum(x for x, _ in tim)
        kern_steps = sum(k for _, k in tim)
        n_launch = kern_steps if mode == "per_step" else len(tim)
        launch_s = kern_ms / 1e3 / n_launch
        bytes_per_launch = bytes_env_step * Eg * (kern_steps / n_launch)   # (the timed launches are those of env group 0)
        achieved = bytes_per_launch / launch_s / 1e9
        traffic, traffic_source = measured_traffic(args.workload, mode, kern_steps / n_launch, Eg)
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_source,
                "kernel": launched_kernel(eng), "avg_launch_us": launch_s * 1e6, "steps_per_launch": kern_steps / n_launch,
                "algorithmic_bytes_per_env_step": bytes_env_step}

    def actor_kernel_times():
        """Policy-in-the-loop runs chain two kernels per step.  Their durations are measured apart: the single-step env kernel by
        HIP events around a train of single-step launches fed by the actor's action buffer, the actor forward as the average
        of a back-to-back train of forwards (stream order: no host gap between them)."""
        loop.reset()
        n = min(T, 64)
        eng.step_n(n, None, 0, None, 0, loop.rew, 0, loop.done, 0, loop.mask, 0, auto_reset=False, persistent=False)
        step_us = eng.last_step_n_kernel_ms() * 1e3 / n
        actor_us = actor.forward_train_us(200) if hasattr(actor, "forward_train_us") else None
        loop.reset()
        return step_us, actor_us

    if args.only_timed:
        roof = {m: None for m in modes}
    elif actor is None:
        roof = {m: roofline(m) for m in modes}
    else:   # no roofline fraction for a two-kernel chain: per-kernel times instead (never a frac computed from mixed durations)
        step_us, actor_us = actor_kernel_times()
        step_frac = bytes_env_step * Eg / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBPS
        roof = {m: None for m in modes}
        actor_times = {"step_kernel_us": step_us, "actor_kernel_us": actor_us, "step_kernel": eng.kernel_name,
                       "step_kernel_roofline_frac": step_frac, "ms_per_step_wall": wall[best] / args.steps * 1e3}
    for e_ in engines:
        e_.check_faults()

    # the configs[4]-shaped number for the driver's default run: a short policy-in-the-loop pass (fused actor obs->400->300->P
    # between single-step launches) on a second handle over the same scenario pool, after the headline measurement
    rollout = None
    if actor is None and not stub and not args.no_rollout_record and not args.only_timed and args.workload != "cfg4":
        rollout = rollout_record(Engine, batch, rk, sk, local_rank, devx, E, wl["lo"], rank, args, T, bytes_env_step)
        try:
            rollout["collector"] = collector_record(Engine, batch, rk, sk, local_rank, devx, E, wl["lo"], rank, args, T)
        except Exception as ex:   # (a record next to the headline: never lets the line fail)
            rollout["collector"] = {"error": str(ex)}
        if sk != _abi.STATE_KINDS["PublicPST"]:   # the same loop with the FLOAT32 policy (two bf16 terms per weight, three per activation: what an SB3 float32 actor computes to 1e-5)
            try:
                r32 = rollout_record(Engine, batch, rk, sk, local_rank, devx, E, wl["lo"], rank, args, T, bytes_env_step, precision="fp32")
                rollout["fp32"] = {k_: r32[k_] for k_ in ("env_steps_per_s_per_gpu", "us_per_step_wall", "precision", "fused_launch", "launches_per_segment",
                                                            "segment_kernel_us_per_step", "actor_kernel_us", "actor", "steps_timed")}
            except Exception as ex:
                rollout["fp32"] = {"error": f"{type(ex).__name__}: {ex}"}

    # outside the timed regions: the C-ABI's own RCCL gather (ev2g_comm_init / ev2g_gather_stats, the path of hosts without
    # torch.distributed) next to torch's, on the same statistics
    c_gather = None
    if multi and hasattr(Engine, "comm_unique_id"):
        # every rank takes part in every collective below whatever fails locally (an exception on one rank only would leave the
        # others waiting): local failures are carried in `err` and agreed on before the next collective step
        from ev2gym_amd.dist import gather_stats_tensor
        err, ids = None, [None]
        if rank == 0:
            try:
                ids = [Engine.comm_unique_id()]
            except Exception as ex:
                err = f"{type(ex).__name__}: {ex}"
        dist.broadcast_object_list(ids, src=0)
        st_all = torch.empty((world * eng.E, _abi.N_STATS), dtype=torch.float64, device=dev)
        st_loc = torch.empty((eng.E, _abi.N_STATS), dtype=torch.float64, device=dev)
        if ids[0] is not None:
            try:
                eng.comm_init(ids[0], rank, world)
            except Exception as ex:
                err = f"{type(ex).__name__}: {ex}"
        ok = torch.tensor([0 if (err or ids[0] is None) else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # the C-ABI collective is entered by all ranks or by none
        if int(ok.item()):
            try:
                eng.gather_stats(st_all)
                eng.stats(out=st_loc)
            except Exception as ex:
                err = f"{type(ex).__name__}: {ex}"
        elif err is None:
            err = "another rank could not initialise its communicator"
        want = gather_stats_tensor(st_loc)
        devx.synchronize()
        if err is None and ids[0] is not None:
            c_gather = {"ranks": eng.comm_world_size, "rows": int(st_all.shape[0]),
                        "equals_torch_all_gather": bool(torch.equal(torch.nan_to_num(st_all, nan=-7.0), torch.nan_to_num(want, nan=-7.0)))}
        else:   # reported, never fatal for the measurement
            c_gather = {"error": err or "rank 0 could not create a communicator id"}

    # what makes an N > 1 line self-verifying (RCCL with more than one rank has never run on hardware available to the build): every rank
    # reports the communicator it is in -- the C-ABI communicator's rank count, a hash of the unique id it was handed -- and a block of GLOBAL
    # env ids travels through the same all-gather as the statistics: block r must hold rank r's ids, first to last
    self_check = None
    if multi:
        import hashlib
        env_ids = torch.arange(E, dtype=torch.int64, device=dev) + rank * E
        all_ids = torch.empty((world * E,), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_ids, env_ids)
        firsts, lasts = [int(all_ids[r * E].item()) for r in range(world)], [int(all_ids[(r + 1) * E - 1].item()) for r in range(world)]
        mine = {"rank": rank, "local_rank": local_rank, "device": str(dev), "c_abi_comm_world_size": (eng.comm_world_size if hasattr(eng, "comm_world_size") else None),
                "unique_id_sha1": (hashlib.sha1(ids[0]).hexdigest()[:12] if (c_gather is not None and ids[0] is not None) else None),
                "first_env": rank * E, "last_env": (rank + 1) * E - 1}
        per = [None] * world
        dist.all_gather_object(per, mine)
        sys.stderr.write(f"[bench rank {rank}] {json.dumps(mine)}\n")
        self_check = {"per_rank": per, "gathered_first_env_of_block": firsts, "gathered_last_env_of_block": lasts,
                      "rank_major_order_ok": firsts == [r * E for r in range(world)] and lasts == [(r + 1) * E - 1 for r in range(world)],
                      "one_unique_id_everywhere": len({p_["unique_id_sha1"] for p_ in per}) == 1,
                      "every_rank_sees_world": all(p_["c_abi_comm_world_size"] in (None, world) for p_ in per)}

    refill = None
    if actor is None and not stub and not args.no_rollout_record and not args.only_timed and args.workload != "cfg4":
        refill = refill_record(Engine, wl, rk, sk, local_rank, devx, E, rank, args, T)

    # the persistent launch with every output KEPT ([T,E,*] blocks), and the other BASELINE single-GPU configs, on the same line
    strided, others = None, None
    if actor is None and not stub and not args.only_timed and n_groups == 1 and "persistent" in modes:
        try:
            strided = strided_record(eng, loop, E, P, D, T, dev, bytes_env_step, (roof.get("persistent") or {}).get("avg_launch_us"))
        except Exception as ex:   # (a record next to the headline: never lets the line fail)
            strided = {"error": f"{type(ex).__name__}: {ex}"}
    if actor is None and not stub and not args.only_timed and not args.no_other_workloads and args.workload == "cfg2" and world == 1:   # (N = 1 records, like cpu_baseline)
        others = {}
        for name in ("cfg3", "cfg4"):
            Eo = max(64, WORKLOADS[name]["envs"] * E // wl["envs"])
            try:
                others[name] = other_workload_record(Engine, name, Eo, local_rank, devx, rank, args)
            except Exception as ex:
                others[name] = {"error": f"{type(ex).__name__}: {ex}"}

    env_steps_total = world * E * args.steps
    value = env_steps_total / wall[best]
    per_rank, ranks_seen = None, None
    if multi:   # what every rank measured itself (the driver computes scaling efficiency from `value`)
        mine = torch.tensor([E * args.steps / float(np.median(res[best][3]))], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [float(v.item()) for v in allv]
        # what the communicator itself reports (not WORLD_SIZE echoed): its size, the blocks that came back from the all-gather
        # above, the rows the last asynchronous statistics gather delivered, and the C-ABI communicator's own rank count
        last = gath.finish()
        ranks_seen = {"process_group_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "all_gather_blocks": len(per_rank), "stats_gather_rows": (int(last.shape[0]) if last is not None else 0),
                      "stats_gather_ranks": (int(last.shape[0]) // E if last is not None else 0),
                      "c_abi_comm_ranks": (c_gather or {}).get("ranks")}
    out = {
        "metric": "env-steps/sec (envs x chargers x steps); % HBM roofline",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall[best] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "reps": res[best][1], "regions_per_rep": res[best][2], "timing": "median over reps of (chain of `regions_per_rep` x `steps`-step regions) / regions_per_rep",
        "config": {"workload": f"{args.workload}: {wl['desc']}", "envs_per_gpu": E, "scenario_pool_per_gpu": M, "scenario_pool_order": "generated" if args.keep_pool_order else "sorted by busy window inside every window of envs_per_gpu scenarios", "chargers": C_,
                   "transformers": R_, "steps_per_episode": T, "obs_dim": D, "occupancy_phi": round(phi, 4),
                   "algorithmic_bytes_per_env_step": bytes_env_step, "soc_log": not args.no_soc_log,
                   "launch": best, "actor": args.actor if actor is None else actor.describe,
                   "actor_env_groups": (n_groups if actor is not None else None),
                   "parallelism": f"env-sharded x{world}, RCCL all_gather of episode stats only (asynchronous; submitted behind the next episode's step kernel: dist.AsyncStatsGather)",
                   "stats_gather_mode": (getattr(gath, "mode", None) if gath is not None else None)},
        "port_steps_per_s": value * P,
        "wall_s_by_launch_mode": {m: round(w, 6) for m, w in wall.items()},
        "env_steps_per_s_by_launch_mode": {m: env_steps_total / w for m, w in wall.items()},
        "full_episode": full_ep,
        "roofline": roof[best],
        "roofline_by_launch_mode": roof,
        "rccl_ranks_seen": ranks_seen, "per_rank_env_steps_per_s": per_rank,
        "rccl_collectives_issued": (gath.collectives if gath is not None else 0),
        "c_abi_rccl_gather": c_gather,
        "rccl_self_check": self_check,
    }
    if actor is not None and not args.only_timed:
        out["actor_kernel_times"] = actor_times
    if rollout is not None:
        out["rollout"] = rollout
    if refill is not None:
        out["device_refill"] = refill
    if strided is not None:
        out["persistent_strided"] = strided
    if others is not None:
        out["other_workloads"] = others
    if stub:
        out["data"] = "STUB ENGINE on CPU (launcher / control-flow test, not a measurement)"
    for e_ in engines:
        e_.close()
    if multi:
        gath.finish()
        dist.barrier()
        dist.destroy_process_group()
    # after the process group is gone (no rank waits for this): the CPU baseline on rank 0's host cores.  At N > 1 the other
    # ranks' processes are winding down meanwhile, so the sample is shorter there and the N = 1 line is the one to quote.
    if rank == 0 and not args.no_cpu_baseline:
        budget = float(os.environ.get("EV2G_BENCH_CPU_BUDGET", "12.0" if world == 1 else "4.0"))
        out["cpu_baseline"] = cpu_baseline(batch.select(np.arange(min(E, 512))), rk, sk, wl["lo"], budget_s=budget)
    elif rank == 0:
        out["cpu_baseline"] = None
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
