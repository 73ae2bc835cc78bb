#!/bin/bash
# round 6, last session: float32 fused launch -- layer 2 walking k-step, tile slot, term (ks1) with a ring of 8 / 4 vs tile slot first (ks0)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32i; mkdir -p $O
EV2G_LIB=build_variants/libev2g_ks1_r8.so timeout 600 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "float32_policy_equals and 37-50" 2>&1 | tail -2 | tee -a $O/pytest.txt
for L in ks0_r4 ks1_r4 ks1_r8 ks0_r4 ks1_r4 ks1_r8; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
EV2G_LIB=build_variants/libev2g_f32st.so timeout 300 python tools/r6/f32_stamps.py 2>&1 | grep -v amdgpu.ids | head -22 | tee $O/f32_stamps_l2ks.txt
tail -3 $O/err.txt
