#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_round3_gpu.py -x -q -k "device_generated" > $O/pytest_refill.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest_refill.log
