#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/gpu_tests.txt
for w in cfg2 cfg3; do python tools/ab_bench.py --workload $w --reps 30 --pool 4 build_variants/noepf.so ev2gym_amd/libev2g_hip.so build_variants/noepf.so ev2gym_amd/libev2g_hip.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_epf_$w.txt; done
AB_SORT=1 python tools/ab_bench.py --workload cfg2 --reps 30 --pool 4 build_variants/noepf.so ev2gym_amd/libev2g_hip.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_epf_cfg2_sorted.txt
AB_STRIDED=1 python tools/ab_bench.py --workload cfg2 --reps 20 --pool 4 build_variants/noepf.so ev2gym_amd/libev2g_hip.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_epf_cfg2_strided.txt
