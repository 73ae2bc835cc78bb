#!/bin/bash
# round 5, call B: the fused actor + step launch -- parity against the two-kernel chain, then the rollout / collector records
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests/test_round5_gpu.py -q -x > $O/fused_tests.txt 2>&1; echo "fused pytest rc=$?"; tail -25 $O/fused_tests.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "collector or rollout or actor" > $O/gpu_tests_sel.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/gpu_tests_sel.txt
for nf in 0 1; do
  if [ $nf = 1 ]; then export EV2G_NO_FUSED=1; else unset EV2G_NO_FUSED; fi
  timeout 300 python tools/sb3_collect_bench.py cfg2 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/collector_cfg2_nofused$nf.json
  timeout 300 python bench.py --actor mlp --no-cpu-baseline 2> $O/bench_actor_$nf.err | tee $O/bench_actor_nofused$nf.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('actor bench nofused=$nf', d['value'], d['ms_per_step'], d.get('actor_kernel_times'))"
done
