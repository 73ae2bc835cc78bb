#!/bin/bash
# round 3, GPU call D: workgroup shapes of ev2g_step_pipe (4 env + 1 worker wavefronts vs 8 + 2), parity on the new default
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fuzz_gpu.py tests/test_round3_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
export EV2G_DEBUG=1
L="ev2gym_amd/libev2g_hip.so build_variants/pipe_8_2.so ev2gym_amd/libev2g_hip.so@EV2G_KERNEL=wave ev2gym_amd/libev2g_hip.so"
timeout 400 python tools/ab_bench.py --workload cfg2 --reps 30 $L 2>&1 | tee $O/ab_cfg2.txt
timeout 400 python tools/ab_bench.py --workload cfg3 --reps 30 --pool 4 $L 2>&1 | tee $O/ab_cfg3.txt
timeout 300 python tools/pipe_timing.py cfg2 2>&1 | tail -16 | tee $O/pipe_timing_cfg2.txt
