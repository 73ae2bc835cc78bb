#!/bin/bash
# round 6: A/B of ev2g_step_big variants (build_variants/*.so given as arguments) at cfg4 + parity of the in-tree library + its phase cycles
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$1; shift; mkdir -p $O
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_round6_gpu.py -q -x -m gpu -k "general_kernel_specialisation or full_size_specialised or cfg4_full_size" 2>&1 | tail -4 | tee $O/parity.txt
timeout 900 python tools/ab_bench.py --workload cfg4 --reps 8 --pool 2 "$@" 2>&1 | grep -v amdgpu.ids | tee $O/ab_cfg4.txt
EV2G_PT_LIB=build_variants/libev2g_pt.so timeout 600 python tools/phase_timing.py cfg4 2>&1 | grep -v amdgpu.ids | tee $O/phase_cfg4_big.txt
