#!/bin/bash
# round 6, third session: ev2g_get_stats_reset_refill (refill kernel forked next to the statistics + reset launch): parity test, then the bench line's device_refill record with and without the overlap
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_fork1; mkdir -p $O
timeout 600 python -m pytest tests/test_round6_gpu.py -q -m gpu -k "stats_reset_refill" -p no:warnings 2>&1 | tail -15 | cut -c1-300 | tee $O/test.txt
for v in "EV2G_X=0" "EV2G_NO_REFILL_OVERLAP=1" "EV2G_X=0" "EV2G_NO_REFILL_OVERLAP=1"; do
  env $v timeout 600 python bench.py --no-cpu-baseline --no-other-workloads > $O/bench.json 2> $O/bench.err; echo "$v bench rc=$?"
  python - "$v" <<'P' | tee -a $O/refill_overlap.txt
import json, sys
d = json.loads(open("gpurun_out/r6_fork1/bench.json").read().strip().splitlines()[-1])
r = d.get("device_refill") or {}
print(sys.argv[1], "| value", round(d["value"] / 1e6, 1), "M | refill us/window", round(r.get("us_per_window", 0), 1), "| episode without", round(r.get("ms_per_episode_without_refill", 0), 4), "ms, with", round(r.get("ms_per_episode_with_refill", 0), 4), "ms =", round(r.get("env_steps_per_s_with_refill_per_gpu", 0) / 1e6, 1), "M env-steps/s")
P
done
