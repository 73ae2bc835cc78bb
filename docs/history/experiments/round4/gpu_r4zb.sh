#!/bin/bash
# round 4, call ZB: the attached session's record requested by the port's own lane in phase A (L1 warm-up for the battery maths)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zb; mkdir -p $O
V=build_variants
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_head.so $V/r4_warm.so $V/r4_head.so $V/r4_warm.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
