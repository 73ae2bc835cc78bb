#!/bin/bash
# statistics gather vs the persistent step kernel: when is the collective handed to the GPU (one rank, RCCL path forced)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5d; mkdir -p $O
for mode in overlap deferred inline; do
EV2G_GATHER_MODE=$mode EV2G_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-cpu-baseline --no-other-workloads --no-rollout-record --launch persistent > $O/dist_$mode.json 2> $O/dist_$mode.err
python - <<PY
import json
d = json.load(open("$O/dist_$mode.json"))
print("$mode", "value %.1f M" % (d["value"] / 1e6), "step launch %.1f us" % d["roofline"]["avg_launch_us"], "episode %.4f ms" % d["full_episode"]["ms_per_episode"], "collectives", d["rccl_collectives_issued"], d["config"]["stats_gather_mode"])
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-other-workloads --no-rollout-record --launch persistent 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no dist', d['value']/1e6, d['roofline']['avg_launch_us'], d['full_episode']['ms_per_episode'])"
