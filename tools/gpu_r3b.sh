#!/bin/bash
# round 3, GPU call B: the software-pipelined persistent kernel (ev2g_step_pipe): parity suite, then A/B against the wave kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
export EV2G_DEBUG=1
L="ev2gym_amd/libev2g_hip.so ev2gym_amd/libev2g_hip.so@EV2G_KERNEL=wave ev2gym_amd/libev2g_hip.so"
timeout 400 python tools/ab_bench.py --workload cfg2 --reps 30 $L 2>&1 | tee $O/ab_cfg2.txt
timeout 400 python tools/ab_bench.py --workload cfg3 --reps 30 --pool 4 $L 2>&1 | tee $O/ab_cfg3.txt
