#!/bin/bash
# round 4, call Y: actor with eight wavefronts per workgroup (two per SIMD) against four
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4y; mkdir -p $O
for w in 4 8; do echo "EV2G_MLP_WAVES=$w"; EV2G_MLP_WAVES=$w timeout 300 python -m pytest tests/test_actor_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E  " | tail -4; done | tee $O/actor_tests.txt
for rep in 1 2; do for w in 4 8; do echo "EV2G_MLP_WAVES=$w"; EV2G_MLP_WAVES=$w timeout 200 python tools/mlp_time.py 2>&1 | grep -v amdgpu.ids; done; done | tee $O/mlp_time.txt
EV2G_MLP_WAVES=8 EV2G_LIB=$PWD/build_variants/mlpt.so timeout 200 python tools/mlp_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/mlp_stamps.txt
