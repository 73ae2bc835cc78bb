#!/bin/bash
# round 4, call ZE: actor for batches beyond 16 x CUs rows: 32 rows per workgroup (a weight fragment feeds two MFMAs)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4ze; mkdir -p $O
timeout 600 python -m pytest tests/test_actor_gpu.py tests/test_round4_gpu.py -m gpu -q -x -k "actor or mlp or rollout or streaming or collector" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E  " | tail -6 | tee $O/tests.txt
for e in 4096 8192 16384; do for v in 1 0; do echo "rows $e EV2G_MLP_NO_BIG=$v"; if [ $v = 1 ]; then EV2G_MLP_NO_BIG=1 timeout 200 python tools/mlp_time.py $e 2>&1 | grep -v amdgpu.ids; else timeout 200 python tools/mlp_time.py $e 2>&1 | grep -v amdgpu.ids; fi; done; done | tee $O/mlp_time.txt
timeout 200 python tools/sb3_collect_bench.py cfg3 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160 | tee $O/collector_cfg3.txt
