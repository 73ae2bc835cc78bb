#!/bin/bash
# HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory) vs the default, on the launch-bound paths
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/ab; mkdir -p $O
for v in 0 1; do
  export HIP_FORCE_DEV_KERNARG=$v
  python bench.py --actor mlp --steps 224 --warmup 28 --no-cpu-baseline > $O/ka_actor_$v.json 2> $O/ka_actor_$v.err
  python bench.py --steps 224 --warmup 28 --no-cpu-baseline --launch per_step > $O/ka_step_$v.json 2> $O/ka_step_$v.err
  python - $O/ka_actor_$v.json $O/ka_step_$v.json $v <<'P'
import json,sys
for f in sys.argv[1:3]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline_by_launch_mode"]
        print("KERNARG",sys.argv[3],f.split("/")[-1], round(d['value']/1e6,2),"M env-steps/s", {m:round(x['avg_launch_us']/x['steps_per_launch'],2) for m,x in r.items()}, flush=True)
    except Exception as e: print("ERR",f,e)
P
done 2>&1 | tee $O/ab_kernarg.txt
