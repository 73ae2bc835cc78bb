#!/bin/bash
# quick parity subset + A/B timing of whatever library variants sit in build_variants/ (first one = reference digest)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/ab; mkdir -p $O
if [ -n "$PYTEST" ]; then python -m pytest $PYTEST -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -5; fi
python tools/ab_bench.py --reps ${REPS:-16} $(ls build_variants/*.so | sort) 2>&1 | tee $O/ab_$(date +%H%M%S).txt | sed 's/digest \[.*//'
