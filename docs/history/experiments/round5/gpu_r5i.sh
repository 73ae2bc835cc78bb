#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5i; mkdir -p $O
B=ev2gym_amd/libev2g_hip.so; R=build_variants/relaxocc.so
AB_SORT=1 python tools/ab_bench.py --workload cfg2 --reps 24 --pool 4 $B $R $B $R $B $R $B $R 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg2.txt
AB_SORT=1 python tools/ab_bench.py --workload cfg3 --reps 24 --pool 4 $B $R $B $R 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg3.txt
for l in $B $R $B $R; do echo -n "$l collector: "; EV2G_LIB=$PWD/$l timeout 300 python tools/sb3_collect_bench.py cfg2 24 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f M  %.3f us/step' % (d['env_steps_per_s']/1e6, d['us_per_step']))"; done | tee $O/collector.txt
