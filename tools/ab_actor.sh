#!/bin/bash
# A/B of library builds (build_variants/*.so) on the GPU box: step kernel per launch mode (tools/ab_bench.py) and the actor rollout
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/ab; mkdir -p $O
LIBS="$@"
python tools/ab_bench.py --reps 18 $LIBS 2>&1 | tee $O/ab_step.txt
for lib in $LIBS; do
  n=$(basename $lib .so)
  for rep in 1; do
    EV2G_LIB=$PWD/$lib python bench.py --actor mlp --steps 224 --warmup 28 --no-cpu-baseline > $O/actor_$n.json 2> $O/actor_$n.err
    python - $O/actor_$n.json $n <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("ACTOR",sys.argv[2], round(d['value']/1e6,2),"M env-steps/s", round(d['ms_per_step']*1e3,2),"us/step", flush=True)
except Exception as e: print("ERR",e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
  done
done 2>&1 | tee $O/ab_actor.txt
