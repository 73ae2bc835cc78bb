"""`python bench.py --gpus N` as the driver types it (no rendezvous in the environment) must start its own N ranks and print ONE
JSON line.  Run here end to end on CPU: the self-launcher, torch.distributed.run, a gloo process group, the timed regions with
their agreed repetition counts, the per-episode asynchronous statistics gather, the whole-episode pass and the rank-derived
fields -- with the CPU oracle standing in for the HIP engine (tests/bench_stub_engine.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, EV2G_BENCH_CPU_BUDGET="0.3", **(env_extra or {}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--engine", "tests.bench_stub_engine:StubEngine",
           "--envs", "4", "--pool", "3", "--steps", "40", "--warmup", "5", "--min-time", "0.02", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_gpus_2_starts_its_own_ranks_and_reports_what_the_communicator_saw():
    out = run_bench("--gpus", "2")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 40 and out["warmup"] == 5
    seen = out["rccl_ranks_seen"]
    assert seen["process_group_size"] == 2 and seen["all_gather_blocks"] == 2 and seen["backend"] == "gloo"
    assert seen["stats_gather_ranks"] == 2 and seen["stats_gather_rows"] == 2 * 4
    assert len(out["per_rank_env_steps_per_s"]) == 2 and all(v > 0 for v in out["per_rank_env_steps_per_s"])
    # whole-job aggregate: both ranks' env-steps over the slowest rank's time
    assert out["value"] == pytest.approx(2 * 4 * 40 / (out["ms_per_step"] * 40 / 1e3), rel=1e-9)
    assert out["rccl_collectives_issued"] >= out["full_episode"]["episodes"]
    assert out["full_episode"]["env_steps_per_s"] > 0
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"
    assert "STUB" in out["data"]
    for m, r in out["roofline_by_launch_mode"].items():
        assert 0 < r["frac"] < 1 and r["kernel"] == "cpu-oracle-stub"
    # round 5: the line verifies itself -- every rank's view of the communicator, and global env ids through the statistics' all-gather
    sc = out["rccl_self_check"]
    assert sc["rank_major_order_ok"] and sc["one_unique_id_everywhere"] and sc["every_rank_sees_world"]
    assert [p["rank"] for p in sc["per_rank"]] == [0, 1] and sc["gathered_first_env_of_block"] == [0, 4] and sc["gathered_last_env_of_block"] == [3, 7]
    assert all(p["c_abi_comm_world_size"] == 2 and p["unique_id_sha1"] for p in sc["per_rank"])
    cg = out["c_abi_rccl_gather"]   # the C-ABI gather's protocol (restated by the stand-in engine): same rows as torch's gather
    assert cg["ranks"] == 2 and cg["rows"] == 8 and cg["equals_torch_all_gather"]


@pytest.mark.timeout(600)
def test_bench_single_process_line_has_the_contract_fields():
    out = run_bench("--gpus", "1", "--launch", "persistent")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out
    assert out["n_gpus"] == 1 and out["rccl_ranks_seen"] is None and out["vs_baseline"] is None and out["dtype"] == "f64"
    assert out["config"]["launch"] == "persistent" and "workload" in out["config"]


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"], env=env, capture_output=True,
                       text=True, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
