#!/bin/bash
# round 4, call ZG: read-back of seven sums instead of eight
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zg; mkdir -p $O
V=build_variants
timeout 700 python tools/ab_bench.py --workload cfg3 --reps 16 --pool 4 $V/r4_head.so $V/r4_nq7.so $V/r4_head.so $V/r4_nq7.so $V/r4_head.so $V/r4_nq7.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg3.txt
