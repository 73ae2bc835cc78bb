#!/bin/bash
# phase cycles of a prebuilt phase-timing library at several env counts: bash tools/r6/gpu_pt.sh <out> <lib> <workload> <envs...>
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$1; L=$2; W=$3; shift 3; mkdir -p $O
for n in "$@"; do echo "## AB_ENVS=$n"; AB_ENVS=$n EV2G_PT_LIB=$L timeout 300 python tools/phase_timing.py $W 2>&1 | grep -v amdgpu.ids | head -10; done | tee $O/pt_$W.txt
