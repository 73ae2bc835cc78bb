#!/bin/bash
# round 6, third session: refill kernel, observation-table rows 0..T-2 as ONE LDS read per lane and row (in-tree) against the round's earlier loop (rows0)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_rf9; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -k "refill or generator or generated or scenario" -p no:warnings 2>&1 | tail -3 | tee -a $O/refill_ab.txt
for L in build_variants/libev2g_rows0.so build_variants/libev2g_rows1.so ev2gym_amd/libev2g_hip.so build_variants/libev2g_rows1.so ev2gym_amd/libev2g_hip.so; do
  for w in cfg2; do
    echo "## $L $w" | tee -a $O/refill_ab.txt
    EV2G_LIB=$L EV2G_REFILL_STAMPS=1 timeout 200 python tools/refill_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-250 | grep -E "stamps|refill of" | tail -2 | tee -a $O/refill_ab.txt
  done
done
