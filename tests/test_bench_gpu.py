"""bench.py's one-line JSON contract (the driver parses it): a small run on the GPU, keys and types checked."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--workload", "cfg3"], ["--actor", "mlp"]], ids=["cfg2", "cfg3", "actor"])
def test_bench_line_contract(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--envs", "512", "--steps", "20", "--warmup", "5", "--min-time", "0.02"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env={**os.environ, "EV2G_BENCH_CPU_BUDGET": "1.5"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 1e6 and abs(d["value"] - 512 * 20 / (d["ms_per_step"] * 20 / 1e3)) <= 1e-6 * d["value"]
    if extra and extra[0] == "--actor":
        # a step is a chain of two kernels: no roofline fraction from mixed durations -- per-kernel times instead, each plausible
        assert d["roofline"] is None
        k = d["actor_kernel_times"]
        assert 1.0 < k["step_kernel_us"] < 200.0 and 1.0 < k["actor_kernel_us"] < 200.0 and 0.0 < k["step_kernel_roofline_frac"] < 1.0
        return
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    ro = d["rollout"]   # the configs[4]-shaped record rides on the default line
    assert ro["env_steps_per_s_per_gpu"] > 1e6 and 1.0 < ro["step_kernel_us"] < 200.0 and 1.0 < ro["actor_kernel_us"] < 200.0
    # the fused actor + step launch: every fast-path shape whose policy fits its packings (round 6: PublicPST too)
    assert ro["fused_launch"] is True and ro["launches_per_segment"] == 1 and 1.0 < ro["segment_kernel_us_per_step"] < 200.0
    assert ro["collector"]["step_kernel_specialisation"] == 4
    if "fp32" in ro:   # (head-table states) the same loop with the float32 policy: fused too since the last session of round 6
        assert ro["fp32"]["precision"] == "fp32" and ro["fp32"]["fused_launch"] is True and ro["fp32"]["env_steps_per_s_per_gpu"] > 1e6, ro["fp32"]
    if not extra:
        ps = d["persistent_strided"]   # every output kept: the strided wide instantiation, with its own roofline record
        assert ps["specialisation"] == 3 and 0.0 < ps["roofline"]["frac"] < 1.0
        ow = d["other_workloads"]      # BASELINE configs[2], configs[3] on the driver's line
        for name, kern in (("cfg3", "ev2g_step_wave<1,1>"), ("cfg4", "ev2g_step_big<512>")):
            assert ow[name]["value"] > 1e5 and 0.0 < ow[name]["roofline"]["frac"] < 1.0 and ow[name]["roofline"]["kernel"].startswith(kern), ow[name]
    rf = d["device_refill"]   # scenario generation on the device: never truncated, cheaper than the episode it feeds
    assert rf["truncated_scenarios"] == 0 and rf["scenarios_per_window"] == 512 and 0.0 < rf["us_per_window"] < 1e4
    assert rf["ms_per_episode_with_refill"] >= 0.5 * rf["ms_per_episode_without_refill"] > 0.0
    if not extra or extra[0] == "--workload":
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb


def test_default_bench_line_holds_the_roofline_fractions():
    """Performance guard (round 6): the default `python bench.py` line -- BASELINE configs[1] at full size, with configs[2] / configs[3] riding on it --
    must keep the roofline fractions the kernels were tuned to.  The fast-path instantiations sit at the 128-register limit and depend on
    `-mllvm -disable-machine-licm` (ev2gym_amd/build.py); the big-env kernel on two workgroups per CU: a toolchain bump that breaks either shows
    up here as a red test instead of a silently slower library.  Floors are ~10 % under the round's measurements (boxes differ by ~3 %)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["config"]["envs_per_gpu"] == 4096 and d["value"] >= 5e6      # the north star's throughput bar, with two orders of magnitude to spare
    assert d["roofline"]["frac"] >= 0.55, d["roofline"]                     # cfg2, outputs overwritten in place (measured 0.60-0.64)
    assert d["persistent_strided"]["roofline"]["frac"] >= 0.48, d["persistent_strided"]   # cfg2, every output kept (0.52-0.55)
    ow = d["other_workloads"]
    assert ow["cfg3"]["roofline"]["frac"] >= 0.32, ow["cfg3"]               # 0.35
    assert ow["cfg4"]["roofline"]["frac"] >= 0.52, ow["cfg4"]               # ev2g_step_big: 0.58-0.62 (ev2g_step_v2<1024, 1>: 0.41)
    assert ow["cfg4"]["roofline"]["kernel"] == "ev2g_step_big<512>" and ow["cfg4"]["specialisation"] == 5
    # the device-resident collector at cfg3: the fused actor + step launch with two PublicPST envs per wavefront (one round of 256 workgroups): 727-737 M measured,
    # 497 M with one env per wavefront, 409-431 M as two launches per step
    col = ow["cfg3"]["collector"]
    assert col["step_kernel_specialisation"] == 4 and col["env_steps_per_s_per_gpu"] >= 6.0e8, col
    # the policy in the loop at cfg2 (BASELINE configs[4]'s per-GPU shard): bf16 operands 495-505 M measured, the float32 policy inside the launch 313-319 M (168-171 M as two launches per step)
    ro = d["rollout"]
    assert ro["fused_launch"] is True and ro["env_steps_per_s_per_gpu"] >= 4.2e8, ro
    assert ro["fp32"]["fused_launch"] is True and ro["fp32"]["env_steps_per_s_per_gpu"] >= 2.6e8, ro["fp32"]
