#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_round3_gpu.py -x -q -k "device_refill or device_generated" > $O/pytest_refill.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_refill.log
EV2G_REFILL_STAMPS=1 timeout 300 python tools/refill_time.py cfg2 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/refill_time.txt
bash tools/gpu_prof_refill.sh cfg2 2>&1 | grep "KT |" | head -4 | tee -a $O/refill_time.txt
bash tools/gpu_prof_refill.sh cfg3 2>&1 | grep "KT |" | head -2 | tee -a $O/refill_time.txt
