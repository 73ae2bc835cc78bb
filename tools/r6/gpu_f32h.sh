#!/bin/bash
# round 6, last session: float32 fused launch -- layer 2 through a deeper ring (r2_N) with / without its LDS read-ahead (aN)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32h; mkdir -p $O
EV2G_LIB=build_variants/libev2g_r2_8_a0.so timeout 600 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "float32_policy_equals and 37-50" 2>&1 | tail -2 | tee -a $O/pytest.txt
for L in r2_4_a1 r2_4_a0 r2_8_a0 r2_10_a0 r2_8_a1 r2_4_a1 r2_4_a0 r2_8_a0 r2_10_a0 r2_8_a1; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
EV2G_LIB=build_variants/libev2g_f32st.so timeout 300 python tools/r6/f32_stamps.py 2>&1 | grep -v amdgpu.ids | head -22 | tee $O/f32_stamps_ring2.txt
tail -3 $O/err.txt
