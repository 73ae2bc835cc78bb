#!/bin/bash
# round 6, final kernels: the randomised sweep under other seed offsets (one line per offset)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_fuzz; mkdir -p $O
for off in $(seq ${FUZZ_FROM:-21} ${FUZZ_TO:-28}); do
  r=$(EV2G_FUZZ_OFFSET=$off timeout 600 python -m pytest tests/test_fuzz_gpu.py -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -1)
  echo "EV2G_FUZZ_OFFSET=$off: $r" | tee -a $O/fuzz.txt
done
