#!/bin/bash
# round 6: refill kernel, the session half and the transformer half in opposite orders by workgroup parity (rfo) against one order (rfo1: same source, -DEV2G_RF_ONE_ORDER) and the tree before (in-tree library)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_rf6; mkdir -p $O
for L in ev2gym_amd/libev2g_hip.so build_variants/libev2g_rfo1.so build_variants/libev2g_rfo.so ev2gym_amd/libev2g_hip.so build_variants/libev2g_rfo1.so build_variants/libev2g_rfo.so; do
  for w in cfg2 cfg3; do
    echo "## $L $w" | tee -a $O/refill_ab.txt
    EV2G_LIB=$L timeout 200 python tools/refill_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tail -1 | tee -a $O/refill_ab.txt
  done
done
EV2G_LIB=build_variants/libev2g_rfo.so timeout 900 python -m pytest tests -q -x -m gpu -k "refill or generat or device_generated or fuzz" 2>&1 | grep -E "passed|failed" | tail -3 | tee $O/parity.txt
