#!/bin/bash
# round 6, last session: float32 fused launch -- layer 3 through a ring of its own (0 = shared ring, 8, 10, 12 slots) x the output layer's tanh (0 = library tanhf, 1 = exp2 + IEEE division)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32f; mkdir -p $O
for L in r3_0_t0 r3_0_t1 r3_8_t0 r3_8_t1 r3_10_t1 r3_12_t1 r3_0_t0 r3_0_t1 r3_8_t0 r3_8_t1 r3_10_t1 r3_12_t1; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
EV2G_LIB=build_variants/libev2g_f32st.so timeout 300 python tools/r6/f32_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/f32_stamps_ring3.txt
tail -3 $O/err.txt
