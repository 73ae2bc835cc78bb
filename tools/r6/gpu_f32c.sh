#!/bin/bash
# round 6, last session: float32 fused launch -- the k-step's five MFMAs as one back-to-back group (G1), + the next k-step's LDS reads ahead of it (G2), ring depths
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32c; mkdir -p $O
for L in g0r8 g1r8 g1r6 g2r6 g2r4 g0r8 g1r8 g1r6 g2r6 g2r4; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
tail -3 $O/err.txt
