#!/bin/bash
# round 6, last session: bf16 fused launch -- two-tile wavefronts walking k-step, tile slot in layer 1 (bit 0) / layer 2 (bit 1) of the inline policy
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_kso; mkdir -p $O
EV2G_LIB=build_variants/libev2g_kso3.so timeout 600 python -m pytest tests/test_round5_gpu.py -x -q -m gpu -k "fused_actor_and_step_launch_equals and 37-50" 2>&1 | tail -2 | tee -a $O/pytest.txt
for L in kso0 kso1 kso2 kso3 kso0 kso1 kso2 kso3; do
  echo "## $L" | tee -a $O/rollout_bf16.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_bf16.txt
done
STAMP_PRECISION=bf16 EV2G_LIB=build_variants/libev2g_f32st.so timeout 300 python tools/r6/f32_stamps.py 2>&1 | grep -v amdgpu.ids | head -22 | tee $O/bf16_stamps_kso3.txt
tail -3 $O/err.txt
