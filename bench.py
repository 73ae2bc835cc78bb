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
from ev2gym_amd.scenario_gen import GenConfig, generate, occupancy_fraction  # noqa: E402

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


def kernel_name(batch, sk, rk):
    """Which step kernel ev2g_load_scenarios() selects for this shape (ev2gym_amd/csrc/ev2g_host.hip)."""
    P, R, npc = batch.n_ports, batch.n_transformers, batch.ports_per_charger
    k = os.environ.get("EV2G_KERNEL", "")
    if P <= 64 and R == 1 and npc == 1 and k == "pipe":
        return f"ev2g_step_pipe<{sk},{rk}>"
    if P <= 64 and R == 1 and npc == 1 and k != "v2":
        return (f"ev2g_step_list<{sk},{rk},{256 if k == 'list256' else 128}>" if k in ("list", "list256") and P >= 4
                else f"ev2g_step_wave<{sk},{rk}>")
    return f"ev2g_step_v2<{256 if P <= 256 else 512 if P <= 512 else 1024}>" if P <= 1024 else "ev2g_step_kernel"


def measured_traffic(workload, launch, steps_per_launch, envs):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC passes (profiles/r01_hbm_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, FETCH doubled per the gfx950 note in
    MI355X_MICROARCH.md), or None when this workload / launch shape was not profiled."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))[workload][launch]
    except Exception:
        return None
    if abs(t["steps_per_launch"] - steps_per_launch) > 1e-9 or envs != WORKLOADS[workload]["envs"]:
        return None
    return (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0


def cpu_baseline(batch, rk, sk, lo, budget_s=12.0):
    """The C oracle (a scalar port of the reference step(), oracle/ev2g_oracle.c) timed on this box's host
    cores, single thread, on whole episodes of a prefix of the same env batch."""
    from oracle.oracle import Oracle
    from ev2gym_amd.engine import host_uniform
    n = min(batch.n_envs, 512)
    sub = batch.select(np.arange(n))
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
    all_cores = dict(value=mt_steps / mt_total, cores=nthr, episodes=mt_eps) if mt_eps else None
    ora.close()
    return dict(value=steps / t_total, unit="env-steps/s", cores=1, kind="port", all_cores=all_cores,
                sample=f"{n} envs x {episodes} episodes x {T} steps of the same workload, C oracle -O2, 1 thread; "
                       f"reference CPython step() measured at build time: 1075 env-steps/s/core at 50 chargers (BASELINE.md)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1120)
    ap.add_argument("--warmup", type=int, default=112)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the workload's)")
    ap.add_argument("--launch", default="auto", choices=["auto", "per_step", "persistent"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-soc-log", action="store_true", help="skip the SoC log that the battery-degradation statistics need")
    ap.add_argument("--actor", default="none", choices=["none", "mlp"],
                    help="mlp: BASELINE configs[4]-shaped rollout -- a torch actor (obs->400->300->P, tanh; SB3-DDPG shape, "
                         "random weights, fp32) produces the actions on the device between steps (forces per_step launches)")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    if args.actor != "none":
        args.launch = "per_step"

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ev2gym_amd.engine import Engine
    wl = WORKLOADS[args.workload]
    E = args.envs or wl["envs"]            # per GPU: weak scaling
    rk, sk = _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]]
    batch = generate(wl["gen"](E, args.seed * 1000 + rank))   # every rank draws its own shard of scenarios
    phi = occupancy_fraction(batch)
    # engine kernels, torch allocations and the RCCL gather all run on ONE explicit (non-default) stream
    tstream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(tstream)
    eng = Engine(batch, rk, sk, device=local_rank, stream=tstream.cuda_stream,
                 flags=0 if args.no_soc_log else _abi.FLAG_LOG_SOC)
    P, D, T = eng.P, eng.D, eng.T
    dev = torch.device("cuda", local_rank)
    acts = torch.empty((T, E, P), dtype=torch.float64, device=dev)
    eng.fill_uniform(acts, T * E * P, 999 + rank, wl["lo"], 1.0)
    obs = torch.empty((E, D), dtype=torch.float64, device=dev)
    rew = torch.empty((E,), dtype=torch.float64, device=dev)
    done = torch.empty((E,), dtype=torch.uint8, device=dev)
    mask = torch.empty((E, P), dtype=torch.uint8, device=dev)
    stats = torch.empty((E, _abi.N_STATS), dtype=torch.float64, device=dev)
    gath = None
    if world > 1:   # double-buffered asynchronous all-gather: the statistics travel while the next episode steps
        from ev2gym_amd.dist import AsyncStatsGather
        gath = AsyncStatsGather(E, world, dev)

    actor = None
    if args.actor == "mlp":
        torch.manual_seed(1234 + rank)
        lo_a = wl["lo"]
        net = torch.nn.Sequential(torch.nn.Linear(D, 400), torch.nn.ReLU(), torch.nn.Linear(400, 300), torch.nn.ReLU(),
                                  torch.nn.Linear(300, P), torch.nn.Tanh()).to(dev)
        a_buf = torch.empty((E, P), dtype=torch.float64, device=dev)

        @torch.no_grad()
        def actor(o):
            a = net(o.to(torch.float32))
            if lo_a == 0.0:
                a = a * 0.5 + 0.5
            a_buf.copy_(a)
            return a_buf

    def run(n_steps, persistent, timing=None):
        """n_steps batched steps; whole episodes where possible; stats (+gather) and reset at episode ends."""
        left = n_steps
        while left > 0:
            t = eng.current_step
            k = 1 if actor is not None else min(left, T - t)
            a_src = actor(obs) if actor is not None else acts[t]
            eng.step_n(k, a_src, E * P, obs, 0, rew, 0, done, 0, mask, 0, auto_reset=False, persistent=persistent)
            if timing is not None:
                timing.append((eng.last_step_n_kernel_ms(), k))
            left -= k
            if eng.current_step >= T:
                if gath is None:
                    eng.stats(out=stats)
                else:
                    eng.stats(out=gath.buffer())
                    gath.launch()
                eng.reset(obs)

    def barrier():
        if gath is not None:
            gath.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(persistent):
        eng.reset(obs)
        run(args.warmup, persistent)
        eng.reset(obs)
        barrier()
        t0 = time.perf_counter()
        run(args.steps, persistent)
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_

    modes = ["per_step", "persistent"] if args.launch == "auto" else [args.launch]
    wall = {m: timed(m == "persistent") for m in modes}
    best = min(wall, key=wall.get)
    # kernel-only duration of the chosen mode, HIP events on the launch stream (separate, untimed pass)
    eng.reset(obs)
    tim = []
    run(min(args.steps, 2 * T), best == "persistent", timing=tim)
    torch.cuda.synchronize()
    kern_ms = sum(x for x, _ in tim)
    kern_steps = sum(k for _, k in tim)
    n_launch = kern_steps if best == "per_step" else len(tim)
    eng.check_faults()

    C_, R_ = batch.n_chargers, batch.n_transformers
    bytes_env_step = P * (phi * wl["b_occ"] + (1 - phi) * wl["b_empty"]) + R_ * wl["b_tr"] + wl["b_env"]
    env_steps_total = world * E * args.steps
    value = env_steps_total / wall[best]
    launch_s = kern_ms / 1e3 / n_launch
    bytes_per_launch = bytes_env_step * E * (kern_steps / n_launch)
    achieved = bytes_per_launch / launch_s / 1e9
    out = {
        "metric": "env-steps/sec (envs x chargers x steps); % HBM roofline",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall[best] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['desc']}", "envs_per_gpu": E, "chargers": C_,
                   "transformers": R_, "steps_per_episode": T, "obs_dim": D, "occupancy_phi": round(phi, 4), "soc_log": not args.no_soc_log,
                   "launch": best, "actor": args.actor, "parallelism": f"env-sharded x{world}, RCCL all_gather of episode stats only (asynchronous, overlaps the next episode)"},
        "port_steps_per_s": value * P,
        "wall_s_by_launch_mode": {m: round(w, 6) for m, w in wall.items()},
        "env_steps_per_s_by_launch_mode": {m: env_steps_total / w for m, w in wall.items()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(args.workload, best, kern_steps / n_launch, E),
                     "kernel": kernel_name(batch, sk, rk),
                     "avg_launch_us": launch_s * 1e6,
                     "steps_per_launch": kern_steps / n_launch,
                     "algorithmic_bytes_per_env_step": bytes_env_step},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(batch, rk, sk, wl["lo"])
    elif rank == 0:
        out["cpu_baseline"] = None
    eng.close()
    if world > 1:
        gath.finish()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
