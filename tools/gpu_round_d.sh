#!/bin/bash
# GPU round D: general-kernel (ev2g_step_v2) work -- parity suite + cfg4 phase timing + cfg4/cfg3 bench lines.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2d; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail
python tools/phase_timing.py cfg4 2>&1 | tail -18 | tee $O/phase_cfg4.txt
python bench.py --workload cfg4 --steps 224 --warmup 28 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg4.json
python - $O/bench_cfg4.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print({m:(round(r['frac'],4),round(r['avg_launch_us']/r['steps_per_launch'],2)) for m,r in d['roofline_by_launch_mode'].items()}, round(d['value']/1e6,2))
P
