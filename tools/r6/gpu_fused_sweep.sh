#!/bin/bash
# round 6, last session: the randomised fused-launch sweep (bf16 and float32 policies against the two-launch chain, bit for bit) under other seed offsets (one line per offset)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_fused_sweep; mkdir -p $O
for off in $(seq ${SWEEP_FROM:-1} ${SWEEP_TO:-12}); do
  r=$(EV2G_FUSED_SWEEP_OFFSET=$off timeout 900 python -m pytest tests/test_round5_gpu.py -q -m gpu -k "randomised_shapes" 2>&1 | grep -E "passed|failed" | tail -1)
  echo "EV2G_FUSED_SWEEP_OFFSET=$off: $r" | tee -a $O/sweep.txt
done
