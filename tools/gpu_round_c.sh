#!/bin/bash
# GPU round C: parity suite after replay-write / topology / RCCL entry points.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2c; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail; grep -B5 -A40 "^___" $O/pytest.log | head -150
