#!/bin/bash
# round 4, call A: parity suite on the new record layout / reciprocal divisions, A/B of the step-kernel variants, phase timing
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4a; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt
V=build_variants
for w in cfg2 cfg3; do timeout 500 python tools/ab_bench.py --workload $w --reps 24 --pool 4 $V/r4_base.so $V/r4_fdiv.so $V/r4_pftop.so $V/r4_rlsum.so $V/r4_base.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
timeout 300 python tools/ab_bench.py --workload cfg4 --reps 6 --pool 2 $V/r4_base.so $V/r4_rlsum.so 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_cfg4.txt
EV2G_PT_LIB=$V/pt_bsplit.so timeout 200 python tools/phase_timing.py cfg2 -DEV2G_PT_BSPLIT 2>&1 | grep -v amdgpu.ids | tee $O/phase_cfg2_bsplit.txt
EV2G_PT_LIB=$V/pt_plain.so timeout 200 python tools/phase_timing.py cfg2 2>&1 | grep -v amdgpu.ids | tee $O/phase_cfg2.txt
EV2G_PT_LIB=$V/pt_outer.so timeout 200 python tools/phase_timing.py cfg2 --outer 2>&1 | grep -v amdgpu.ids | tee $O/phase_cfg2_outer.txt
