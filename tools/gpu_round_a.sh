#!/bin/bash
# GPU round A: parity suite, driver-shaped and default bench lines, cfg3/cfg4/actor lines, profiles, phase timing.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r2a; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 1500 $O/bench_driver.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --workload cfg4 --steps 224 --warmup 28 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --actor mlp --steps 224 --warmup 28 --no-cpu-baseline > $O/bench_actor.json 2> $O/bench_actor.err
python bench.py --pool 1 --no-cpu-baseline > $O/bench_pool1.json 2> $O/bench_pool1.err
for f in default cfg3 cfg4 actor pool1; do echo "== $f"; python - $O/bench_$f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','reps','wall_s_by_launch_mode','full_episode')})
    print({m:(round(r['frac'],4),round(r['avg_launch_us'],2)) for m,r in d['roofline_by_launch_mode'].items()}, d.get('cpu_baseline') and round(d['cpu_baseline']['value']))
except Exception as e: print('ERR',e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
done
bash tools/prof_step.sh cfg2_persistent --launch persistent > $O/prof_cfg2_persistent.txt 2>&1
python tools/phase_timing.py cfg2 > $O/phase_cfg2.txt 2>&1
python tools/phase_timing.py cfg2 --outer > $O/phase_cfg2_outer.txt 2>&1
tail -30 $O/phase_cfg2.txt
