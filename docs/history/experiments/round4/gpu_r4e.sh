#!/bin/bash
# round 4, call E: statistics kernel restructured (session-window test, one round trip less per session) + fused statistics/reset launch
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4e; mkdir -p $O
V=build_variants
timeout 600 python -m pytest tests/test_round4_gpu.py -q -x 2>&1 | tail -15 | tee $O/gpu_tests_r4.txt
for l in r4_hist r4_stats; do EV2G_LIB=$PWD/$V/$l.so timeout 200 python tools/stats_time.py cfg2 cfg3 cfg4 2>&1 | grep -v amdgpu.ids | tee -a $O/stats_time.txt; done
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|Error|error" $O/gpu_tests.txt | tail -15
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cat $O/bench_default.json | cut -c1-1500
