#!/bin/bash
# Round-2 evidence at HEAD: parity suite, smoke, bench lines, rocprofv3 kernel-trace + PMC summaries, HBM-traffic JSON, phase timings.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2f; rm -rf $O; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed|real" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_driver_shaped.json 2> $O/bench_driver.err
python bench.py > $O/bench_cfg2_default.json 2> $O/bench_default.err
python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --workload cfg4 --steps 224 --warmup 28 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --actor mlp --steps 224 --warmup 28 --no-cpu-baseline > $O/bench_cfg2_actor_mlp.json 2> $O/bench_actor.err
python bench.py --actor mlp_torch --steps 112 --warmup 28 --no-cpu-baseline > $O/bench_cfg2_actor_torch.json 2> $O/bench_actor_torch.err
for f in cfg2_driver_shaped cfg2_default cfg3 cfg4 cfg2_actor_mlp cfg2_actor_torch; do echo "== $f"; python - $O/bench_$f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','reps')}, d.get('full_episode') and round(d['full_episode']['env_steps_per_s']/1e6,1))
    print({m:(round(r['frac'],4),round(r['avg_launch_us']/r['steps_per_launch'],2)) for m,r in d['roofline_by_launch_mode'].items()}, d.get('cpu_baseline') and round(d['cpu_baseline']['value']))
except Exception as e: print('ERR',e)
P
done
bash tools/prof_step.sh cfg2_persistent --launch persistent > $O/r02_cfg2_persistent_rocprofv3.txt 2>&1
bash tools/prof_step.sh cfg2_per_step --launch per_step > $O/r02_cfg2_per_step_rocprofv3.txt 2>&1
bash tools/prof_step.sh cfg3_persistent --workload cfg3 --launch persistent > $O/r02_cfg3_persistent_rocprofv3.txt 2>&1
bash tools/prof_step.sh cfg4_persistent --workload cfg4 --steps 224 --warmup 28 --launch persistent > $O/r02_cfg4_persistent_rocprofv3.txt 2>&1
python tools/collect_evidence.py $O/r02_hbm_traffic.json cfg2_persistent=cfg2:persistent cfg2_per_step=cfg2:per_step cfg3_persistent=cfg3:persistent cfg4_persistent=cfg4:persistent > $O/collect.log 2>&1; tail -3 $O/collect.log
rm -rf gpurun_out/prof_cfg2_persistent gpurun_out/prof_cfg2_per_step gpurun_out/prof_cfg3_persistent gpurun_out/prof_cfg4_persistent
python tools/phase_timing.py cfg2 2>&1 | grep -v amdgpu.ids > $O/r02_phase_cfg2.txt
python tools/phase_timing.py cfg4 2>&1 | grep -v amdgpu.ids > $O/r02_phase_cfg4.txt
python tools/occupancy_probe.py 2>&1 | grep -v amdgpu.ids > $O/r02_occupancy_probe.txt
grep "^KT" $O/r02_*_rocprofv3.txt | head -20
