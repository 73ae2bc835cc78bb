#!/bin/bash
# round 6, last session: float32 fused launch -- input rows split into their bf16 terms once per row at the policy's entry (xs1; a1 = layer 1 reads a k-step ahead) vs per operand fragment (xs0); parity first
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32g; mkdir -p $O
for L in xs1a0 xs1a1; do
EV2G_LIB=build_variants/libev2g_$L.so timeout 600 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "float32_policy_equals and V2G_profit_max_loads" 2>&1 | tail -3 | tee -a $O/pytest.txt
done
for L in xs0 xs1a1 xs1a0 xs1a0r6 xs0 xs1a1 xs1a0 xs1a0r6; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
EV2G_LIB=build_variants/libev2g_f32st.so timeout 300 python tools/r6/f32_stamps.py 2>&1 | grep -v amdgpu.ids | head -22 | tee $O/f32_stamps_xsplit.txt
tail -3 $O/err.txt
