#!/bin/bash
# round 4, call ZC: the env's episode accumulators read together with the reduction's operands (phase D) instead of at the start of phase E
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zc; mkdir -p $O
V=build_variants
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_head.so $V/r4_eapre1.so $V/r4_eapre2.so $V/r4_head.so $V/r4_eapre1.so $V/r4_eapre2.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
