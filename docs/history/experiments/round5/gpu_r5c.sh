#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5c; mkdir -p $O
timeout 600 python -m pytest tests/test_round5_gpu.py -q -x > $O/fused_tests.txt 2>&1; echo "fused pytest rc=$?"; tail -5 $O/fused_tests.txt
EV2G_PT_LIB=build_variants/pt_fused.so timeout 300 python tools/phase_timing_fused.py 2>&1 | grep -v amdgpu.ids | tail -9 | tee $O/phase_fused.txt
timeout 300 python tools/sb3_collect_bench.py cfg2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
timeout 300 python bench.py --actor mlp --no-cpu-baseline 2> $O/bench_actor.err | tee $O/bench_actor.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('actor bench', d['value'], d['ms_per_step'])"
