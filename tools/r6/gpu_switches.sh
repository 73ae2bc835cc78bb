#!/bin/bash
# round 6, final tree: the engine / surface suites under the library's A/B switches (every alternative path still agrees with the oracle and with the default path)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_switches; mkdir -p $O
for v in "EV2G_X=0" "EV2G_NO_DICT=1" "EV2G_NO_FUSED=1" "EV2G_FUSED_ONE_ENV=1" "EV2G_NO_BIG=1" "EV2G_KERNEL=v2" "EV2G_NO_FULL=1" "EV2G_NO_WIDE=1" "EV2G_NO_STRIDED=1" "EV2G_ROLLOUT_GRAPHS=0"; do
  r=$(env $v timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fuzz_gpu.py tests/test_actor_gpu.py tests/test_python_surface_gpu.py tests/test_round4_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1)
  echo "$v: $r" | tee -a $O/switches.txt
done
