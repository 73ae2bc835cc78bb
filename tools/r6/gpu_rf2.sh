#!/bin/bash
# round 6: parts of the refill kernel's setpoint phase (library built with -DEV2G_RF_SUBSTAMPS), timing of the final variant, parity
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_rf4; mkdir -p $O
for w in cfg2 cfg3; do
  echo "## $w substamps" | tee -a $O/sub.txt
  EV2G_LIB=build_variants/libev2g_rf4.so EV2G_REFILL_STAMPS=1 timeout 200 python tools/refill_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-400 | grep -E "parts|stamps|refill of" | tail -5 | tee -a $O/sub.txt
done
for L in build_variants/libev2g_rf0.so build_variants/libev2g_rf5.so build_variants/libev2g_rf0.so build_variants/libev2g_rf5.so; do
  for w in cfg2 cfg3; do
    echo "## $L $w" | tee -a $O/refill_ab.txt
    EV2G_LIB=$L timeout 200 python tools/refill_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tail -1 | tee -a $O/refill_ab.txt
  done
done
EV2G_LIB=build_variants/libev2g_rf5.so timeout 900 python -m pytest tests -q -x -m gpu -k "refill or generat or device_generated or fuzz" 2>&1 | tail -3 | tee $O/parity.txt
