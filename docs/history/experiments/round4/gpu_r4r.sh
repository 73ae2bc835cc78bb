#!/bin/bash
# round 4, call R: float32 actor as split bf16 operands on the streaming kernel (2 / 3 terms per weight) against the float32-MFMA kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4r; mkdir -p $O
for t in 3 2; do echo "EV2G_MLP_F32_TERMS=$t"; EV2G_MLP_F32_TERMS=$t timeout 600 python -m pytest tests/test_actor_gpu.py tests/test_round3_gpu.py -m gpu -q -x -k "actor or mlp or rollout" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E  " | tail -8; done | tee $O/actor_tests.txt
for t in 3 2; do echo "EV2G_MLP_F32_TERMS=$t"; EV2G_MLP_F32_TERMS=$t MLP_PREC=fp32 timeout 200 python tools/mlp_time.py 2>&1 | grep -v amdgpu.ids; done | tee $O/mlp_time.txt
echo "EV2G_MLP_OLD=1"; EV2G_MLP_OLD=1 MLP_PREC=fp32 timeout 200 python tools/mlp_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/mlp_time.txt
timeout 200 python tools/mlp_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/mlp_time.txt
