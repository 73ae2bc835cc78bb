#!/bin/bash
# round 6, last session: float32 fused launch -- weight requests through a scalar base (the wavefront's number read as a scalar: scalar tile guards, `global_load v, v_lane16, s[base]`): sa1 vs sa0; sa1r6 = with a shared ring of 6
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32k; mkdir -p $O
EV2G_LIB=build_variants/libev2g_sa1.so timeout 600 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "float32_policy_equals and 37-50" 2>&1 | tail -2 | tee -a $O/pytest.txt
for L in sa0 sa1 sa1r6 sa0 sa1 sa1r6; do
  echo "## $L" | tee -a $O/rollout_fp32.txt
  EV2G_LIB=build_variants/libev2g_$L.so timeout 300 python bench.py --actor mlp_fp32 --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/rollout_fp32.txt
done
tail -3 $O/err.txt
