#!/bin/bash
# round 3, GPU call A: parity suite, driver-shaped bench line (with the new rollout record), A/B of the prepared LDS-batching patches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; cut -c1-600 $O/bench_driver.json
L="ev2gym_amd/libev2g_hip.so build_variants/lds_batch_C.so build_variants/lds_batch_A.so build_variants/lds_batch_AC.so ev2gym_amd/libev2g_hip.so"
timeout 400 python tools/ab_bench.py --workload cfg2 --reps 30 $L 2>&1 | tee $O/ab_cfg2.txt
timeout 400 python tools/ab_bench.py --workload cfg3 --reps 30 --pool 4 $L 2>&1 | tee $O/ab_cfg3.txt
