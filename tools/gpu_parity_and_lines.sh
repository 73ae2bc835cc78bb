#!/bin/bash
# GPU round G: parity suite after fusing the remaining reward built-ins + quick cfg2 / cfg4 lines (no regression check).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2g; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail; grep -B3 -A25 "^___" $O/pytest.log | head -80
for w in "cfg2" "cfg3" "cfg4 --steps 224 --warmup 28"; do set -- $w
python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$1.json
python - $O/bench_$1.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print(d['config']['workload'][:5], {m:(round(r['frac'],4),round(r['avg_launch_us']/r['steps_per_launch'],2)) for m,r in d['roofline_by_launch_mode'].items()}, round(d['value']/1e6,2), d['roofline']['traffic'])
P
done
python bench.py --actor mlp --steps 224 --warmup 28 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_actor.json
python - $O/bench_actor.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print("actor rollout", round(d['value']/1e6,2), "M env-steps/s", round(d['ms_per_step']*1e3,2), "us/step")
P
