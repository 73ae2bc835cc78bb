#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4i; mkdir -p $O
V=build_variants
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 24 --pool 4 $V/r4_head.so $V/x_rec0.so $V/r4_head.so $V/x_rec0.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
timeout 600 python -m pytest tests/test_round4_gpu.py -q -x 2>&1 | tail -5 | tee $O/gpu_tests_r4.txt
