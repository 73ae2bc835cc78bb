#!/bin/bash
# residency census of ev2g_step_pipe (4 + 1 wavefronts per workgroup): time per step against the number of envs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3e; mkdir -p $O
for E in 2048 3072 3584 4096; do
  echo "== envs $E"; AB_ENVS=$E timeout 300 python tools/ab_bench.py --workload cfg2 --reps 12 --pool 2 ev2gym_amd/libev2g_hip.so ev2gym_amd/libev2g_hip.so@EV2G_KERNEL=wave 2>&1 | grep persistent
done | tee $O/census.txt
