#!/bin/bash
# round 6: refill kernel, the lane-per-step loops of prices / spawner tables / transformer series unrolled by two (rfu) against the tree (in-tree library)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_rf7; mkdir -p $O
for L in ev2gym_amd/libev2g_hip.so build_variants/libev2g_rfu.so ev2gym_amd/libev2g_hip.so build_variants/libev2g_rfu.so; do
  for w in cfg2 cfg3; do
    echo "## $L $w" | tee -a $O/refill_ab.txt
    EV2G_LIB=$L EV2G_REFILL_STAMPS=1 timeout 200 python tools/refill_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-250 | grep -E "stamps|refill of" | tail -2 | tee -a $O/refill_ab.txt
  done
done
EV2G_LIB=build_variants/libev2g_rfu.so timeout 900 python -m pytest tests -q -x -m gpu -k "refill or generat or device_generated" 2>&1 | grep -E "passed|failed" | tail -3 | tee $O/parity.txt
