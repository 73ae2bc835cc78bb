#!/bin/bash
# round 4, call J: session records staged in LDS (FULLK = 3): parity suite + A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4j; mkdir -p $O
V=build_variants
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/gpu_tests.txt | tail -15
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 24 --pool 4 $V/r4_head.so $V/r4_stg.so $V/r4_head.so $V/r4_stg.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
