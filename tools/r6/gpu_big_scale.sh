#!/bin/bash
# how two resident workgroups share a CU: ev2g_step_big with 256 (1 per CU), 512 (2 per CU), 1024, 2048 envs; and ev2g_step_v2<1024,1> with 256, 512
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$1; shift; mkdir -p $O
for n in 256 512 1024 2048; do echo "## AB_ENVS=$n"; AB_ENVS=$n timeout 600 python tools/ab_bench.py --workload cfg4 --reps 8 --pool 2 "$@" 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//'; done | tee $O/scale_cfg4.txt
