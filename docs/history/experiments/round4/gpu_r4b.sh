#!/bin/bash
# round 4, call B: parity suite on the per-port state line / occupancy masks, A/B against the previous variants
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|Error|error" $O/gpu_tests.txt | tail -15
V=build_variants
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 24 --pool 4 $V/r4_base.so $V/r4_rlsum.so $V/r4_line.so $V/r4_rlsum.so $V/r4_line.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
timeout 300 python tools/ab_bench.py --workload cfg4 --reps 6 --pool 2 $V/r4_rlsum.so $V/r4_line.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg4.txt
