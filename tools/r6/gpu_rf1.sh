#!/bin/bash
# round 6: ev2g_refill_kernel -- spawn trials with the lanes re-converged for the session draws (pass 1) and packed (session, step-of-stay) pairs for the power setpoints.
# A/B against the serial variants (build_variants/libev2g_rf0.so = -DEV2G_RF_SERIAL_PASS1 -DEV2G_RF_SERIAL_SETPOINTS), parity of every refill / generator test under the new library
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_rf2; mkdir -p $O
for L in build_variants/libev2g_rf0.so build_variants/libev2g_rf2.so build_variants/libev2g_rf3.so; do
  for w in cfg2 cfg3; do
    echo "## $L $w" | tee -a $O/refill_ab.txt
    EV2G_LIB=$L EV2G_REFILL_STAMPS=1 timeout 200 python tools/refill_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -6 | tee -a $O/refill_ab.txt
  done
done
EV2G_LIB=build_variants/libev2g_rf3.so timeout 900 python -m pytest tests -q -x -m gpu -k "refill or generat or device_generated or fuzz" 2>&1 | tail -5 | tee $O/parity.txt
