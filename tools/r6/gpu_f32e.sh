#!/bin/bash
# round 6, last session: float32 fused launch incl. PublicPST -- parity (new + the precision-parametrised sweeps), cfg3 / cfg2 rollout rates fused vs two launches
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32e; mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py tests/test_actor_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
for W in cfg3 cfg2; do for NF in 0 1; do
  echo "## $W EV2G_NO_FUSED_F32=$NF" | tee -a $O/rollout_fp32.txt
  if [ $NF = 1 ]; then export EV2G_NO_FUSED_F32=1; else unset EV2G_NO_FUSED_F32; fi
  timeout 300 python bench.py --workload $W --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done; done
tail -3 $O/err.txt
