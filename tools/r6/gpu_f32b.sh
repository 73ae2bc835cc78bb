#!/bin/bash
# round 6, last session: float32 fused launch -- which wavefronts take layer 3's tiles (0..3 / 4..7 / 12..15), ONLY_00 builds; stamps of the 4..7 form
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32b; mkdir -p $O
for L in build_variants/libev2g_w3off0.so build_variants/libev2g_w3off4.so build_variants/libev2g_w3off12.so build_variants/libev2g_w3off0.so build_variants/libev2g_w3off4.so build_variants/libev2g_w3off12.so; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=$L timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
EV2G_LIB=build_variants/libev2g_f32st.so timeout 300 python tools/r6/f32_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/f32_stamps_w3off4.txt
tail -3 $O/err.txt
