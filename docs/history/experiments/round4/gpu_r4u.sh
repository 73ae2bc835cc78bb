#!/bin/bash
# round 4, call U: ranked SoC log, masks of a batch read before its loads
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4u; mkdir -p $O
V=build_variants
for l in r4_head r4_rank r4_rank2; do EV2G_LIB=$PWD/$V/$l.so timeout 200 python tools/stats_time.py cfg2 cfg3 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v "eng.reset" | tee -a $O/stats_time.txt; done
