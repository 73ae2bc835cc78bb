#!/bin/bash
# round 6, last session: the float32 policy inside the fused actor + step launch -- parity tests, then the rollout rate under ring depths 6 / 8 (in-tree) / 10 / 12
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32a; mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "float32" 2>&1 | tail -15 | tee $O/pytest_new.txt
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_actor_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_regress.txt
for L in ev2gym_amd/libev2g_hip.so build_variants/libev2g_ringf6.so build_variants/libev2g_ringf10.so build_variants/libev2g_ringf12.so ev2gym_amd/libev2g_hip.so; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=$L timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('actor_kernel_times'))" | tee -a $O/rollout_fp32.txt
done
echo "## EV2G_NO_FUSED_F32=1" | tee -a $O/rollout_fp32.txt
EV2G_NO_FUSED_F32=1 timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 2>>$O/err.txt | tail -1 | cut -c1-400 | tee -a $O/rollout_fp32.txt
tail -5 $O/err.txt
