#!/bin/bash
# last refresh of the cfg2 lines after the final statistics / reset changes: default bench line, driver-shaped line, persistent-launch profile (with PMC passes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4ev2; mkdir -p $O/summaries
timeout 300 python bench.py > $O/r04_bench_cfg2_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_cfg2_driver_shaped.json 2> $O/bench_driver.err; echo "driver rc=$?"
bash tools/prof_step.sh cfg2_persistent --workload cfg2 --launch persistent > $O/r04_cfg2_persistent_rocprofv3.txt 2>&1; tail -8 $O/r04_cfg2_persistent_rocprofv3.txt
cp gpurun_out/prof_cfg2_persistent/summary.json $O/summaries/cfg2_persistent.json; rm -rf gpurun_out/prof_*
for w in cfg2 cfg3 cfg4; do timeout 200 python tools/stats_time.py $w 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/r04_stats_time.txt
