#!/bin/bash
# round 4, call ZH: the envs sums side by side in LDS, read back with 16-byte loads
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zh; mkdir -p $O
V=build_variants
timeout 700 python tools/ab_bench.py --workload cfg3 --reps 16 --pool 4 $V/r4_head.so $V/r4_esums.so $V/r4_head.so $V/r4_esums.so $V/r4_head.so $V/r4_esums.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg3.txt
timeout 500 python tools/ab_bench.py --workload cfg2 --reps 16 --pool 4 $V/r4_head.so $V/r4_esums.so $V/r4_head.so $V/r4_esums.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg2.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E  " | tail -4
