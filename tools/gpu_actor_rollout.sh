#!/bin/bash
# actor-in-the-loop rollout: HIP-graph replay of rollout segments; one chain vs pipelined env groups
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests/test_actor_gpu.py -q 2>&1 | tail -3
for cfg in "1 1" "1 0" "2 1" "4 1"; do set -- $cfg
  EV2G_ROLLOUT_GRAPHS=$2 python bench.py --actor mlp --actor-groups $1 --steps 224 --warmup 28 --no-cpu-baseline > $O/actor_g$1_graph$2.json 2> $O/actor_g$1_graph$2.err
  python - $O/actor_g$1_graph$2.json $1 $2 <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("groups",sys.argv[2],"graphs",sys.argv[3], round(d['value']/1e6,2),"M env-steps/s", round(d['ms_per_step']*1e3,2),"us/step")
except Exception as e: print("ERR",e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
done
