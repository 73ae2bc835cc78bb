#!/bin/bash
# round 4, call ZF: several envs per wavefront: the env's episode accumulators read with the reduction's read-back (one LDS round trip less)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zf; mkdir -p $O
V=build_variants
timeout 700 python tools/ab_bench.py --workload cfg3 --reps 16 --pool 4 $V/r4_head.so $V/r4_eb.so $V/r4_head.so $V/r4_eb.so $V/r4_head.so $V/r4_eb.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg3.txt
