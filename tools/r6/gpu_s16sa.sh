#!/bin/bash
# round 6, last session: the stand-alone streaming actor (ev2g_mlp3_s16) with scalar weight bases (s16sa1) vs vector addresses (s16sa0): forwards back to back, three precisions; two-launch rollout
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_s16sa; mkdir -p $O
EV2G_LIB=build_variants/libev2g_s16sa1.so timeout 600 python -m pytest tests/test_round4_gpu.py tests/test_actor_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a $O/pytest.txt
for L in s16sa0 s16sa1 s16sa0 s16sa1; do
  echo "## $L" | tee -a $O/mlp_time.txt
  for P in bf16 fp32 fp32x3; do MLP_PREC=$P EV2G_LIB=build_variants/libev2g_$L.so timeout 200 python tools/mlp_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/mlp_time.txt; done
done
