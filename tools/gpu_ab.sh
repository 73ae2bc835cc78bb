#!/bin/bash
# A/B of library variants on the GPU box: bash tools/gpu_ab.sh <out dir under gpurun_out> <lib> <lib> ...
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$1; shift; mkdir -p $O
for w in cfg2 cfg3; do python tools/ab_bench.py --workload $w --reps 30 --pool 4 "$@" 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
