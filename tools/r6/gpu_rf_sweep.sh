#!/bin/bash
# round 6, third session: the randomised device-refill sweep (device == host generator by behaviour) under other seed offsets on the restructured row loop; then the default bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_rf_sweep; mkdir -p $O
for off in 0 31 32 33 34 35 36 37 38; do
  r=$(EV2G_FUZZ_OFFSET=$off timeout 600 python -m pytest tests/test_round6_gpu.py tests/test_round3_gpu.py -q -m gpu -k "refill or generated" -p no:warnings 2>&1 | grep -E "passed|failed" | tail -1)
  echo "EV2G_FUZZ_OFFSET=$off: $r" | tee -a $O/sweep.txt
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d = json.loads(open("gpurun_out/r6_rf_sweep/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "full_episode", d.get("full_episode"))
print("device_refill", d.get("device_refill"))
P
