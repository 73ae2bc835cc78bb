#!/bin/bash
# round 4, call S: actor precision modes (tests) + single-step launch with the session record requested in the prologue
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4s; mkdir -p $O
timeout 600 python -m pytest tests/test_actor_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py -m gpu -q -x -k "actor or mlp or rollout or float32 or collector" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E  " | tail -8 | tee $O/actor_tests.txt
for p in bf16 fp32 fp32x3; do MLP_PREC=$p timeout 200 python tools/mlp_time.py 2>&1 | grep -v amdgpu.ids; done | tee $O/mlp_time.txt
V=build_variants
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_head.so $V/r4_warm.so $V/r4_head.so $V/r4_warm.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
