#!/bin/bash
# round 4, call ZI: statistics kernel, SoC-log pass with all lanes at the same step (coalesced rows) against the session-relative loop
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zi; mkdir -p $O
V=build_variants
for l in r4_head r4_tsweep r4_head r4_tsweep; do EV2G_LIB=$PWD/$V/$l.so timeout 300 python tools/stats_time.py cfg2 cfg3 cfg4 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v "eng.reset" | tee -a $O/stats_time.txt; done
