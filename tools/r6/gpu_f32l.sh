#!/bin/bash
# round 6, last session: float32 fused launch -- raised issue priority (s_setprio 3) for the wavefronts that hold two tiles in layer 1 (bit 0) / layer 2 (bit 1)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32l; mkdir -p $O
for L in prio0 prio1 prio2 prio3 prio0 prio1 prio2 prio3; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
tail -3 $O/err.txt
