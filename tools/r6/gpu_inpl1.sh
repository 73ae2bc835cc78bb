#!/bin/bash
# round 6, third session: battery maths IN PLACE (ev2g_step_wave's INPL instantiations, EV2G_INPLACE=1) against the compacted worker list: cfg3, cfg2, strided; parity under the switch
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_inpl1; mkdir -p $O
L=ev2gym_amd/libev2g_hip.so
python tools/ab_bench.py --workload cfg3 --reps 30 --pool 4 $L "$L@EV2G_INPLACE=1" $L "$L@EV2G_INPLACE=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_cfg3.txt
python tools/ab_bench.py --workload cfg2 --reps 20 --pool 4 $L "$L@EV2G_INPLACE=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_cfg2.txt
echo "## EV2G_INPLACE=1" | tee -a $O/parity.txt
EV2G_INPLACE=1 timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fuzz_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_round5_gpu.py -q -m gpu -p no:warnings 2>&1 | tail -8 | cut -c1-300 | tee -a $O/parity.txt
