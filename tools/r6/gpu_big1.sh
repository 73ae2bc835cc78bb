#!/bin/bash
# round 6, call 2: the big-env kernel (ev2g_step_big): parity first, then A/B against ev2g_step_v2<1024, 1>, then its phase cycles
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_big1; mkdir -p $O
L=ev2gym_amd/libev2g_hip.so
timeout 900 python -m pytest tests/test_round3_gpu.py -q -x -m gpu -k "general_kernel_specialisation or full_size_specialised" 2>&1 | tail -15 | tee $O/parity_spec.txt
timeout 1200 python -m pytest tests/test_round6_gpu.py -q -x -m gpu 2>&1 | tail -15 | tee $O/parity_r6.txt
timeout 600 python tools/ab_bench.py --workload cfg4 --reps 8 --pool 2 "$L@EV2G_NO_BIG=1" $L 2>&1 | grep -v amdgpu.ids | tee $O/ab_cfg4.txt
EV2G_PT_LIB=build_variants/libev2g_pt.so timeout 600 python tools/phase_timing.py cfg4 2>&1 | grep -v amdgpu.ids | tee $O/phase_cfg4_big.txt
EV2G_NO_BIG=1 EV2G_PT_LIB=build_variants/libev2g_pt.so timeout 600 python tools/phase_timing.py cfg4 2>&1 | grep -v amdgpu.ids | tee $O/phase_cfg4_v2.txt
