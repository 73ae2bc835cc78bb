#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for l in f_base f_e6p3 f_e6p4 f_e6p3r9 f_base f_e6p3; do
  echo -n "$l: "; EV2G_LIB=$PWD/build_variants/$l.so timeout 300 python tools/sb3_collect_bench.py cfg2 24 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f M  %.3f us/step  spec %d' % (d['env_steps_per_s']/1e6, d['us_per_step'], d['specialisation']))"
done
