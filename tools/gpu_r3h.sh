#!/bin/bash
# A/B of the FULL-kernel variants + parity run on the current library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3h
L="build_variants/full_v5.so build_variants/full_v6.so build_variants/full_v7.so build_variants/full_v3.so"
python tools/ab_bench.py --workload cfg2 --reps 30 --pool 4 $L > gpurun_out/r3h/ab_cfg2.txt 2>&1
python tools/ab_bench.py --workload cfg3 --reps 30 --pool 4 $L > gpurun_out/r3h/ab_cfg3.txt 2>&1
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fuzz_gpu.py tests/test_round3_gpu.py -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3h/pytest.log; cat gpurun_out/r3h/ab_cfg2.txt gpurun_out/r3h/ab_cfg3.txt
