#!/bin/bash
# round 3, GPU call F: whole parity suite on the current tree (LDS-batched fast path, agents through the facade, float32 actor), bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; cut -c1-400 $O/bench_driver.json
timeout 300 python bench.py --actor mlp_fp32 --no-cpu-baseline > $O/bench_actor_fp32.json 2> $O/bench_actor_fp32.err; echo "bench fp32 rc=$?"; cut -c1-300 $O/bench_actor_fp32.json
timeout 300 python bench.py --actor mlp --no-cpu-baseline > $O/bench_actor_bf16.json 2> $O/bench_actor_bf16.err; echo "bench bf16 rc=$?"; cut -c1-300 $O/bench_actor_bf16.json
