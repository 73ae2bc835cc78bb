#!/bin/bash
# round 6, last session: bf16 fused launch -- weight requests through a scalar base (bsa1) vs vector addresses (bsa0); bsa1r12 = with a ring of 12 fragments
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_bsa; mkdir -p $O
EV2G_LIB=build_variants/libev2g_bsa1.so timeout 600 python -m pytest tests/test_round5_gpu.py -x -q -m gpu -k "fused_actor_and_step_launch_equals and 37-50" 2>&1 | tail -2 | tee -a $O/pytest.txt
for L in bsa0 bsa1 bsa1r12 bsa0 bsa1 bsa1r12; do
  echo "## $L" | tee -a $O/rollout_bf16.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_bf16.txt
done
tail -3 $O/err.txt
