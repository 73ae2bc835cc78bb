#!/bin/bash
# round 6: whole GPU suite + the default bench line on the tree as it stands
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r6_full}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests_full.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests_full.txt | tail -3 | tee $O/gpu_tests.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
print("device_refill", {k: d["device_refill"][k] for k in ("us_per_window", "ms_per_episode_without_refill", "ms_per_episode_with_refill", "env_steps_per_s_with_refill_per_gpu")})
print("full_episode", d["full_episode"]["ms_per_episode"], "launch us", d["roofline"]["avg_launch_us"])
for k, v in d["other_workloads"].items(): print(k, v["value"], v["roofline"]["frac"])
PY
