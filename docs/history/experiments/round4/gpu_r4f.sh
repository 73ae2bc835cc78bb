#!/bin/bash
# round 4, call F: statistics kernel with the SoC-log pass dealt out in chunks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4f; mkdir -p $O
V=build_variants
for l in r4_stats r4_chunk; do EV2G_LIB=$PWD/$V/$l.so timeout 200 python tools/stats_time.py cfg2 cfg3 cfg4 2>&1 | grep -v amdgpu.ids | tee -a $O/stats_time.txt; done
EV2G_STATS_SEQUENTIAL=1 EV2G_LIB=$PWD/$V/r4_chunk.so timeout 200 python tools/stats_time.py cfg2 2>&1 | grep -v amdgpu.ids | tee -a $O/stats_time.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|Error|error" $O/gpu_tests.txt | tail -15
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench_default.json').read())
print(d['value'], d['roofline']['frac'], d['full_episode'], d['roofline']['avg_launch_us'])
PY
