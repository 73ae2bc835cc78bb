#!/bin/bash
# evidence, part B: rocprofv3 kernel-trace summaries (cfg2 both launch modes, cfg3, cfg4, actor rollout) + PMC passes for cfg2 persistent
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2g; mkdir -p $O
bash tools/prof_step.sh cfg2_persistent --launch persistent > $O/r02_cfg2_persistent_rocprofv3.txt 2>&1
PASSES=kt bash tools/prof_step.sh cfg2_per_step --launch per_step > $O/r02_cfg2_per_step_rocprofv3.txt 2>&1
PASSES=kt bash tools/prof_step.sh cfg3_persistent --workload cfg3 --launch persistent > $O/r02_cfg3_persistent_rocprofv3.txt 2>&1
PASSES=kt bash tools/prof_step.sh cfg4_persistent --workload cfg4 --steps 224 --warmup 28 --launch persistent > $O/r02_cfg4_persistent_rocprofv3.txt 2>&1
PASSES=kt bash tools/prof_step.sh cfg2_actor --actor mlp --steps 224 --warmup 28 > $O/r02_cfg2_actor_rocprofv3.txt 2>&1
python tools/collect_evidence.py $O/r02_hbm_traffic.json cfg2_persistent=cfg2:persistent > $O/collect.log 2>&1; tail -3 $O/collect.log
rm -rf gpurun_out/prof_cfg2_persistent gpurun_out/prof_cfg2_per_step gpurun_out/prof_cfg3_persistent gpurun_out/prof_cfg4_persistent gpurun_out/prof_cfg2_actor
grep "^KT" $O/r02_*_rocprofv3.txt | head -24
