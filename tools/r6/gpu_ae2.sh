#!/bin/bash
# round 6: fused actor + step launch with two PublicPST envs per wavefront (32 policy rows per workgroup): parity against the two-kernel chain, collector / rollout A/B
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r6_ae2}; mkdir -p $O
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_actor_gpu.py -q -x -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 | tee $O/parity.txt
for v in "EV2G_FUSED_ONE_ENV=1" "EV2G_X=0" "EV2G_FUSED_ONE_ENV=1" "EV2G_X=0"; do
  echo "## $v" | tee -a $O/collector_cfg3.txt
  env $v timeout 200 python tools/sb3_collect_bench.py cfg3 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160 | tee -a $O/collector_cfg3.txt
done
for v in "EV2G_FUSED_ONE_ENV=1" "EV2G_X=0"; do
  echo "## $v" | tee -a $O/rollout_cfg3.txt
  env $v timeout 300 python bench.py --workload cfg3 --actor mlp --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('rollout'))[:300])" | tee -a $O/rollout_cfg3.txt
done
