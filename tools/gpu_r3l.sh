#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3l
L="build_variants/full_v7.so build_variants/full_v9.so"
python tools/ab_bench.py --workload cfg2 --reps 30 --pool 4 $L > gpurun_out/r3l/ab_cfg2.txt 2>&1
python tools/ab_bench.py --workload cfg3 --reps 30 --pool 4 $L > gpurun_out/r3l/ab_cfg3.txt 2>&1
cat gpurun_out/r3l/ab_cfg2.txt gpurun_out/r3l/ab_cfg3.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3l/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed" gpurun_out/r3l/pytest.log
