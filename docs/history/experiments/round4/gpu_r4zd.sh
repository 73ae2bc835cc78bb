#!/bin/bash
# round 4, call ZD: accumulator reads hoisted (instantiations with room): full GPU suite + A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zd; mkdir -p $O
V=build_variants
timeout 1200 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/gpu_tests.txt | tail -6
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_head.so $V/r4_eapre.so $V/r4_head.so $V/r4_eapre.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
