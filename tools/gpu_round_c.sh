#!/bin/bash
# GPU round C: parity suite after replay-write / topology / RCCL entry points, cfg4 profile for the general-kernel work.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2c; mkdir -p $O
( time python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
bash tools/prof_step.sh cfg4 --workload cfg4 --steps 224 --warmup 28 > $O/prof_cfg4.txt 2>&1; tail -40 $O/prof_cfg4.txt
