#!/bin/bash
# round 4, call O: the 16-row streaming actor kernel (ev2g_mlp3_s16) against round 3's (EV2G_MLP_OLD=1)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4o; mkdir -p $O
timeout 600 python -m pytest tests/test_actor_gpu.py -m gpu -q -x > $O/actor_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/actor_tests.txt; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/actor_tests.txt | tail -12
for v in 0 1; do echo "EV2G_MLP_OLD=$v"; EV2G_MLP_OLD=$v timeout 200 python tools/mlp_time.py 2>&1 | grep -v amdgpu.ids; done | tee $O/mlp_time.txt
for v in 0 1; do echo "EV2G_MLP_OLD=$v"; EV2G_MLP_OLD=$v EV2G_LIB=$PWD/build_variants/mlpt.so timeout 200 python tools/mlp_stamps.py 2>&1 | grep -v amdgpu.ids; done | tee $O/mlp_stamps.txt
