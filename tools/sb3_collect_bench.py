#!/usr/bin/env python3
"""Throughput of the device-resident off-policy collector (sb3_vec_env.DeviceReplayCollector over ev2g_collect) at a bench workload:
whole episodes of every env, actor forward -> env step with the transitions written in place, statistics + reset at the episode ends.
  python tools/sb3_collect_bench.py [cfg2|cfg3] [episodes] [bf16|fp32|fp32x3]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS
from ev2gym_amd import _abi
from ev2gym_amd.actor import init_mlp_weights
from ev2gym_amd.engine import Engine
from ev2gym_amd.sb3_vec_env import DeviceReplayCollector
from ev2gym_amd.scenario_gen import generate_native

wname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n_ep = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"   # the policy's operand precision (DeviceReplayCollector(precision=...))
wl = WORKLOADS[wname]
E = wl["envs"]
batch = generate_native(wl["gen"](4 * E, 0))
eng = Engine(batch, _abi.REWARD_KINDS[wl["reward"]], _abi.STATE_KINDS[wl["state"]], flags=_abi.FLAG_LOG_SOC, n_active_envs=E)
col = DeviceReplayCollector(eng, init_mlp_weights(eng.D, eng.P, seed=1), wl["lo"], capacity_episodes=3, precision=prec, use_torch=False)
col.collect_episode(); eng.synchronize()
t0 = time.perf_counter()
for _ in range(n_ep):
    col.collect_episode()
eng.synchronize()
dt = time.perf_counter() - t0
eng.check_faults()
print(json.dumps({"workload": wname, "policy_precision": prec, "envs": E, "episodes": n_ep, "env_steps_per_s": E * eng.T * n_ep / dt, "us_per_step": dt / (n_ep * eng.T) * 1e6,
                  "specialisation": eng.last_launch_specialisation, "replay_bytes_per_episode_block": int((eng.T + 1) * E * eng.D * 4 + eng.T * E * (eng.P * 5 + 9)),
                  "note": "DeviceReplayCollector: obs / action / reward / done / mask rows written in place by the actor and step kernels (no host copies); "
                          "includes get_statistics + reset at every episode end"}))
