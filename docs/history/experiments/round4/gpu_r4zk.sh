#!/bin/bash
# round 4, call ZK: ev2g_step_v2 hot accesses as base + 32-bit offset (cfg4)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zk; mkdir -p $O
V=build_variants
timeout 800 python tools/ab_bench.py --workload cfg4 --reps 6 --pool 2 $V/r4_head.so $V/r4_v2i32.so $V/r4_head.so $V/r4_v2i32.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg4.txt
