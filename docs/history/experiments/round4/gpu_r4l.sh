#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4l; mkdir -p $O
V=build_variants
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_now.so $V/r4_now.so@AB_SORT=1 $V/r4_now.so $V/r4_now.so@AB_SORT=1 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
timeout 600 python -m pytest tests/test_round4_gpu.py -q 2>&1 | tail -3
