#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6_cfg4ev; mkdir -p $O
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -c 600 $O/bench_default.err
timeout 400 python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
PASS_TIMEOUT=200 bash tools/prof_step.sh cfg4_persistent --workload cfg4 --launch persistent > $O/cfg4_persistent_rocprofv3.txt 2>&1; tail -12 $O/cfg4_persistent_rocprofv3.txt
cp gpurun_out/prof_cfg4_persistent/summary.json $O/cfg4_summary.json; rm -rf gpurun_out/prof_*
timeout 600 python -m pytest tests/test_bench_gpu.py -q -x -m gpu 2>&1 | tail -3
