#!/bin/bash
# round 4, call N: kernel-argument / parameter-block line prefetch at the top of the fast-path kernel (single-step launches)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4n; mkdir -p $O
V=build_variants
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_head.so $V/r4_kpref.so $V/r4_head.so $V/r4_kpref.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
