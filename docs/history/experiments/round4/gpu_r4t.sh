#!/bin/bash
# round 4, call T: ranked SoC log (rows hold the step's entries side by side; the statistics kernel ranks by the scenario's occupancy masks)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4t; mkdir -p $O
V=build_variants
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/gpu_tests.txt | tail -12
for l in r4_head r4_rank; do EV2G_LIB=$PWD/$V/$l.so timeout 200 python tools/stats_time.py cfg2 cfg3 2>&1 | grep -v amdgpu.ids | tee -a $O/stats_time.txt; done
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_head.so $V/r4_rank.so $V/r4_head.so $V/r4_rank.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
