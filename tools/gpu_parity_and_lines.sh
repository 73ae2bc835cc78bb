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
# kernel-trace averages of the step / statistics / reset kernels at cfg2 and cfg3 (persistent launches)
for w in cfg2 cfg3; do rm -rf $O/kt_$w; rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python bench.py --workload $w --launch persistent --no-cpu-baseline --min-time 0.1 > /dev/null 2>&1
python - $O/kt_$w/kt_results.db $w <<'P'
import sqlite3, sys
con=sqlite3.connect(sys.argv[1])
for r in con.execute("select name,total_calls,average from top_kernels limit 3"): print(sys.argv[2], "KT |", r[0][:60], "|", r[1], "|", round(r[2],2))
P
rm -rf $O/kt_$w; done
