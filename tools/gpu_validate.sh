#!/bin/bash
# quick validation on the GPU box: parity suite, smoke, default bench line; results under gpurun_out/val
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/val; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -c 1500 $O/bench_default.json
