#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r4g; mkdir -p $O
V=build_variants
for l in r4_chunk x_nolog x_nohist x_nosess; do EV2G_LIB=$PWD/$V/$l.so timeout 200 python tools/stats_time.py cfg2 2>&1 | grep -v amdgpu.ids | tee -a $O/stats_time.txt; done
cd /tmp; EV2G_LIB=$GRAFT_REPO_ROOT/$V/r4_chunk.so rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/stats_time.py cfg2 > /tmp/kt.log 2>&1; f=$(ls /tmp/kt/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find /tmp/kt -name "*kernel_stats*" | head -1); head -8 "$f" | cut -c1-200 | tee $GRAFT_REPO_ROOT/$O/kt_stats.txt
