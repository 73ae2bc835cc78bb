#!/bin/bash
# round 6, last session: compact SoC-log rows (entry j of a row = the j-th occupied port; envs of 33..64 ports) -- parity, then the statistics kernel and the whole-episode rates against port-indexed rows
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_soc; mkdir -p $O
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_round5_gpu.py tests/test_round6_gpu.py tests/test_actor_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
for L in build_variants/libev2g_soc0.so ev2gym_amd/libev2g_hip.so build_variants/libev2g_soc0.so ev2gym_amd/libev2g_hip.so; do
  echo "## $L" | tee -a $O/ab.txt
  EV2G_LIB=$L timeout 200 python tools/stats_time.py cfg2 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/ab.txt
  EV2G_LIB=$L timeout 300 python bench.py --no-other-workloads --no-cpu-baseline --no-rollout-record 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'launch_us', d['roofline']['avg_launch_us'], 'episode', d['full_episode']['ms_per_episode'])" | tee -a $O/ab.txt
done
tail -3 $O/err.txt
