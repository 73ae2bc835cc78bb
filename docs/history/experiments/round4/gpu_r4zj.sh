#!/bin/bash
# round 4, call ZJ: statistics kernel, SoC-log sums in capacity units (no float64 multiplication per entry)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4zj; mkdir -p $O
V=build_variants
for l in r4_head r4_raw r4_head r4_raw; do EV2G_LIB=$PWD/$V/$l.so timeout 300 python tools/stats_time.py cfg2 cfg3 cfg4 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v "eng.reset" | tee -a $O/stats_time.txt; done
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E  " | tail -6
