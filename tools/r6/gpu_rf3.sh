#!/bin/bash
# round 6: refill kernel, observation tables written row by row (rf6) against the strip loops (rf5) and the round-5 kernel (rf0); parity
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_rf5; mkdir -p $O
for L in build_variants/libev2g_rf0.so build_variants/libev2g_rf5.so build_variants/libev2g_rf6.so build_variants/libev2g_rf5.so build_variants/libev2g_rf6.so; do
  for w in cfg2 cfg3; do
    echo "## $L $w" | tee -a $O/refill_ab.txt
    EV2G_LIB=$L EV2G_REFILL_STAMPS=1 timeout 200 python tools/refill_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-300 | grep -E "stamps|refill of" | tail -2 | tee -a $O/refill_ab.txt
  done
done
EV2G_LIB=build_variants/libev2g_rf6.so timeout 900 python -m pytest tests -q -x -m gpu -k "refill or generat or device_generated or fuzz" 2>&1 | tail -3 | tee $O/parity.txt
