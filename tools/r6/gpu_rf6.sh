#!/bin/bash
# round 6: refill kernel, observation-table rows as streaming (non-temporal) stores (rfnt) against ordinary stores (in-tree)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_rf8; mkdir -p $O
for L in ev2gym_amd/libev2g_hip.so build_variants/libev2g_rfnt.so ev2gym_amd/libev2g_hip.so build_variants/libev2g_rfnt.so; do
  for w in cfg2; do
    echo "## $L $w" | tee -a $O/refill_ab.txt
    EV2G_LIB=$L EV2G_REFILL_STAMPS=1 timeout 200 python tools/refill_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-250 | grep -E "stamps|refill of" | tail -2 | tee -a $O/refill_ab.txt
  done
done
