#!/bin/bash
# round 6, call 1: cfg3 envs-per-wavefront A/B (EV2G_EPW_CAP / EV2G_EPW_ALIGN), parity of every width under the knobs, baseline numbers
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_ab1; mkdir -p $O
L=ev2gym_amd/libev2g_hip.so
python tools/ab_bench.py --workload cfg3 --reps 30 --pool 4 $L "$L@EV2G_EPW_CAP=2" "$L@EV2G_EPW_CAP=2,EV2G_EPW_ALIGN=1" "$L@EV2G_EPW_CAP=1" "$L@EV2G_EPW_ALIGN=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_cfg3_epw.txt
python tools/ab_bench.py --workload cfg2 --reps 20 --pool 4 $L 2>&1 | grep -v amdgpu.ids | tee $O/ab_cfg2.txt
python tools/ab_bench.py --workload cfg4 --reps 6 --pool 2 $L 2>&1 | grep -v amdgpu.ids | tee $O/ab_cfg4.txt
for v in "EV2G_EPW_CAP=2" "EV2G_EPW_CAP=2 EV2G_EPW_ALIGN=1" "EV2G_EPW_CAP=1"; do
  echo "## $v" | tee -a $O/parity.txt
  env $v timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -m gpu -k "every_env_width or golden or batched or persistent_multi" 2>&1 | tail -3 | tee -a $O/parity.txt
done
lscpu | grep -E "^(CPU\(s\)|Thread|Core|Socket|Model name)" | tee $O/lscpu.txt
