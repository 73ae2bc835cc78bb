#!/bin/bash
# round 4, call H: round-4 tests (collector, warnings, fused statistics/reset, masks), whole suite, collector throughput, default bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4h; mkdir -p $O
timeout 600 python -m pytest tests/test_round4_gpu.py -q -x 2>&1 | tail -25 | tee $O/gpu_tests_r4.txt
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|^FAILED|^ERROR" $O/gpu_tests.txt | tail -15
timeout 200 python tools/sb3_collect_bench.py cfg2 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/collect_cfg2.json
timeout 200 python tools/sb3_collect_bench.py cfg3 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/collect_cfg3.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4h/bench_default.json').read())
print(d['value'], d['roofline']['frac'], d['full_episode'], d['roofline']['avg_launch_us'], d['roofline'].get('traffic_source'))
PY
