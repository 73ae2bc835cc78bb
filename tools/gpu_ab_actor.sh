#!/bin/bash
# rollout A/B (fused actor between single-step launches): bash tools/gpu_ab_actor.sh <lib> <lib> ...
cd $GRAFT_REPO_ROOT; O=gpurun_out/ab_actor; mkdir -p $O
for rep in 1 2; do for lib in "$@"; do
  n=$(basename $lib .so)
  EV2G_LIB=$PWD/$lib python bench.py --actor mlp --no-cpu-baseline > $O/actor_$n.json 2> $O/actor_$n.err
  python - $O/actor_$n.json $n <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k=d["actor_kernel_times"]
    print("ACTOR", sys.argv[2], round(d["value"]/1e6,2), "M env-steps/s", round(d["ms_per_step"]*1e3,2), "us/step wall; step kernel", round(k["step_kernel_us"],2), "actor kernel", round(k["actor_kernel_us"],2), flush=True)
except Exception as e: print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
P
done; done 2>&1 | tee $O/ab_actor.txt
