#!/bin/bash
# round 6, last session: float32 fused launch -- the hidden activations' rows 4..11 with the halves of every 32-byte pair swapped (conflict-free ds_read_b128 groups): hswz1 vs hswz0
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32j; mkdir -p $O
EV2G_LIB=build_variants/libev2g_hswz1.so timeout 600 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "float32_policy_equals and 37-50" 2>&1 | tail -2 | tee -a $O/pytest.txt
for L in hswz0 hswz1 hswz0 hswz1 hswz0 hswz1; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
EV2G_LIB=build_variants/libev2g_f32st.so timeout 300 python tools/r6/f32_stamps.py 2>&1 | grep -v amdgpu.ids | head -22 | tee $O/f32_stamps_hswz.txt
tail -3 $O/err.txt
