#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4k; mkdir -p $O
V=build_variants
EV2G_PT_LIB=$V/pt_plain.so timeout 200 python tools/phase_timing.py cfg2 2>&1 | grep -v amdgpu.ids | head -11 | tee $O/phase_cfg2_stg.txt
EV2G_NO_STAGED=1 EV2G_PT_LIB=$V/pt_plain.so timeout 200 python tools/phase_timing.py cfg2 2>&1 | grep -v amdgpu.ids | head -11 | tee $O/phase_cfg2_nostg.txt
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 16 --pool 4 $V/r4_stg.so@EV2G_NO_STAGED=1 $V/r4_stg.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/gpu_tests.txt | tail -8
