#!/bin/bash
# GPU round B: parity of the rewritten fast path, then A/B timing of the library variants.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2b; mkdir -p $O
( time python -m pytest tests -m gpu -q -x ) > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log
python tools/ab_bench.py --reps 24 build_variants/libev2g_base.so build_variants/libev2g_new.so build_variants/libev2g_new3.so build_variants/libev2g_new3_noquiet.so > $O/ab_cfg2.txt 2>&1; cat $O/ab_cfg2.txt
python tools/ab_bench.py --workload cfg3 --reps 12 build_variants/libev2g_base.so build_variants/libev2g_new3.so > $O/ab_cfg3.txt 2>&1; cat $O/ab_cfg3.txt
python tools/phase_timing.py cfg2 > $O/phase_cfg2.txt 2>&1; tail -22 $O/phase_cfg2.txt
