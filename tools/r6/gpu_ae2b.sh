#!/bin/bash
# round 6: weight-ring depth of the two-envs-per-wavefront fused launch (7 in-tree, 10, 13)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_ae2b; mkdir -p $O
for L in ev2gym_amd/libev2g_hip.so build_variants/libev2g_ring10.so build_variants/libev2g_ring13.so ev2gym_amd/libev2g_hip.so build_variants/libev2g_ring10.so build_variants/libev2g_ring13.so; do
  echo "## $L" | tee -a $O/collector_cfg3.txt
  EV2G_LIB=$L timeout 200 python tools/sb3_collect_bench.py cfg3 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160 | tee -a $O/collector_cfg3.txt
done
