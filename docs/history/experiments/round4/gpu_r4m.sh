#!/bin/bash
# round 4, call M: dense SoC log (rank-compacted writes, port-per-lane sweeps in the statistics kernel)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4m; mkdir -p $O
V=build_variants
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/gpu_tests.txt | tail -12
for l in r4_prev r4_clog; do EV2G_LIB=$PWD/$V/$l.so timeout 200 python tools/stats_time.py cfg2 cfg3 2>&1 | grep -v amdgpu.ids | tee -a $O/stats_time.txt; done
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_prev.so $V/r4_clog.so $V/r4_prev.so $V/r4_clog.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
