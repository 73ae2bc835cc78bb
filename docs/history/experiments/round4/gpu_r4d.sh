#!/bin/bash
# round 4, call D: env-major interleaved histories -- statistics kernel time, step kernel A/B, parity suite
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4d; mkdir -p $O
V=build_variants
for l in r4_line r4_hist; do EV2G_LIB=$PWD/$V/$l.so timeout 200 python tools/stats_time.py cfg2 cfg3 cfg4 2>&1 | grep -v amdgpu.ids | tee -a $O/stats_time.txt; done
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 24 --pool 4 $V/r4_line.so $V/r4_hist.so $V/r4_line.so $V/r4_hist.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
timeout 300 python tools/ab_bench.py --workload cfg4 --reps 6 --pool 2 $V/r4_line.so $V/r4_hist.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg4.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|Error|error" $O/gpu_tests.txt | tail -15
