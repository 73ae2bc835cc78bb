#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3i
L="build_variants/full_v7.so build_variants/full_v8.so build_variants/rot_lo.so build_variants/rot_hi.so build_variants/rot_none.so"
python tools/ab_bench.py --workload cfg2 --reps 30 --pool 4 $L > gpurun_out/r3i/ab_cfg2.txt 2>&1
python tools/ab_bench.py --workload cfg3 --reps 30 --pool 4 $L > gpurun_out/r3i/ab_cfg3.txt 2>&1
cat gpurun_out/r3i/ab_cfg2.txt gpurun_out/r3i/ab_cfg3.txt
