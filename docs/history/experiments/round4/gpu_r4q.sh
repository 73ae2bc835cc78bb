#!/bin/bash
# round 4, call Q: actor prologue variants (requests between the conversion steps; HEAD 20 / 28 / 36)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4q; mkdir -p $O
timeout 600 python -m pytest tests/test_actor_gpu.py -m gpu -q -x > $O/actor_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/actor_tests.txt; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/actor_tests.txt | tail -12
for rep in 1 2; do for l in "" build_variants/mlp_h20.so build_variants/mlp_h36.so; do echo "lib=$l"; EV2G_LIB=${l:+$PWD/$l} timeout 200 python tools/mlp_time.py 2>&1 | grep -v amdgpu.ids | head -1; done; done | tee $O/mlp_time.txt
EV2G_LIB=$PWD/build_variants/mlpt.so timeout 200 python tools/mlp_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/mlp_stamps.txt
