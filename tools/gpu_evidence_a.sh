#!/bin/bash
# evidence, part A: parity suite, smoke, bench lines (default / driver-shaped cfg2, cfg3, cfg4, actor rollout)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2g; rm -rf $O; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed|real" $O/pytest.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2_driver_shaped.json 2> $O/bench_driver.err
python bench.py > $O/bench_cfg2_default.json 2> $O/bench_default.err
EV2G_BENCH_CPU_BUDGET=4 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
EV2G_BENCH_CPU_BUDGET=4 python bench.py --workload cfg4 --steps 224 --warmup 28 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --actor mlp --steps 224 --warmup 28 --no-cpu-baseline > $O/bench_cfg2_actor_mlp.json 2> $O/bench_actor.err
for f in cfg2_driver_shaped cfg2_default cfg3 cfg4 cfg2_actor_mlp; do echo "== $f"; python - $O/bench_$f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','reps')}, d.get('full_episode') and round(d['full_episode']['env_steps_per_s']/1e6,1))
    print({m:(round(r['frac'],4),round(r['avg_launch_us']/r['steps_per_launch'],2)) for m,r in d['roofline_by_launch_mode'].items()}, d.get('cpu_baseline') and round(d['cpu_baseline']['value']))
except Exception as e: print('ERR',e)
P
done
