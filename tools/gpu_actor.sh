#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/actor; mkdir -p $O
python -m pytest tests/test_actor_gpu.py -m gpu -q -x 2>&1 | tail -15
for a in mlp mlp_torch; do python bench.py --actor $a --steps 224 --warmup 28 --no-cpu-baseline > $O/bench_$a.json 2> $O/bench_$a.err; python - $O/bench_$a.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['config']['actor'][:60], 'value %.1f M  ms/step %.4f'%(d['value']/1e6, d['ms_per_step']))
except Exception as e: print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
done
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --actor mlp --steps 224 --warmup 28 --no-cpu-baseline --min-time 0.05 > $O/kt.log 2>&1
python - <<'P'
import sqlite3
con=sqlite3.connect('gpurun_out/actor/kt/kt_results.db')
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 6"): print("KT |", r[0][:60], "|", r[1], "|", round(r[2],1), "|", round(r[3],3), "|", round(r[4],2))
P
