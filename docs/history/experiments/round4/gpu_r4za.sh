#!/bin/bash
# round 4, call ZA: per-env reductions of a wavefront's two or three envs side by side (cfg3)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4za; mkdir -p $O
V=build_variants
for w in cfg3 cfg2; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_head.so $V/r4_dpar.so $V/r4_head.so $V/r4_dpar.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
