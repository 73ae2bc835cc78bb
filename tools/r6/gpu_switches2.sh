#!/bin/bash
# round 6: the failing cases of gpu_switches.sh by name and first assertion (are they routing assertions or wrong results?)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_switches; mkdir -p $O
for v in "EV2G_KERNEL=v2" "EV2G_NO_FULL=1" "EV2G_NO_WIDE=1" "EV2G_NO_STRIDED=1" "EV2G_ROLLOUT_GRAPHS=0"; do
  echo "=== $v" >> $O/failures.txt
  env $v timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fuzz_gpu.py tests/test_actor_gpu.py tests/test_python_surface_gpu.py tests/test_round4_gpu.py -q -m gpu --tb=line -p no:warnings 2>&1 | grep -vE "^\.|^$" | cut -c1-400 | head -150 >> $O/failures.txt
done
tail -c 3000 $O/failures.txt
