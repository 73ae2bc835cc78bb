#!/bin/bash
# round 6 evidence (run on the GPU box via gpurun): parity suite, smoke, bench lines, rocprofv3 kernel traces + PMC passes; results under gpurun_out/r6ev
# (.head_sha is written by the caller: `git rev-parse HEAD > .head_sha` -- the box has no .git).  Every rocprofv3 pass runs under a timeout (tools/prof_step.sh).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6ev; mkdir -p $O
echo "source: $(cat .head_sha 2>/dev/null)" | tee $O/r06_gpu_tests.txt
timeout 1500 python -m pytest tests -m gpu -q >> $O/r06_gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/r06_gpu_tests.txt; tail -3 $O/r06_gpu_tests.txt
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/r06_bench_cfg2_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_cfg2_driver_shaped.json 2> $O/bench_driver.err; echo "driver rc=$?"
timeout 300 python bench.py --workload cfg3 > $O/r06_bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
timeout 400 python bench.py --workload cfg4 > $O/r06_bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
timeout 300 python bench.py --actor mlp --no-cpu-baseline > $O/r06_bench_cfg2_actor_mlp.json 2> $O/bench_actor.err; echo "actor rc=$?"
timeout 300 python bench.py --workload cfg3 --actor mlp --no-cpu-baseline > $O/r06_bench_cfg3_actor_mlp.json 2> $O/bench_actor3.err; echo "actor cfg3 rc=$?"
timeout 300 python bench.py --actor mlp_fp32 --no-cpu-baseline > $O/r06_bench_cfg2_actor_mlp_fp32.json 2> $O/bench_actor32.err; echo "actor32 rc=$?"
timeout 300 python bench.py --workload cfg3 --actor mlp_fp32 --no-cpu-baseline > $O/r06_bench_cfg3_actor_mlp_fp32.json 2> $O/bench_actor32_3.err; echo "actor32 cfg3 rc=$?"
EV2G_NO_FUSED_F32=1 timeout 300 python bench.py --actor mlp_fp32 --no-cpu-baseline --no-other-workloads > $O/r06_bench_cfg2_actor_mlp_fp32_two_launches.json 2> $O/bench_actor32_nf.err; echo "actor32 unfused rc=$?"
EV2G_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-cpu-baseline --no-other-workloads > $O/r06_bench_cfg2_torchrun_world1_forced_dist.json 2> $O/bench_dist.err; echo "dist rc=$?"
for spec in "cfg2_persistent --workload cfg2 --launch persistent" "cfg2_per_step --workload cfg2 --launch per_step" "cfg3_persistent --workload cfg3 --launch persistent" "cfg4_persistent --workload cfg4 --launch persistent"; do
  set -- $spec; tag=$1; shift
  PASS_TIMEOUT=240 bash tools/prof_step.sh $tag "$@" > $O/r06_${tag}_rocprofv3.txt 2>&1; tail -8 $O/r06_${tag}_rocprofv3.txt
  mkdir -p $O/summaries; cp gpurun_out/prof_$tag/summary.json $O/summaries/r06_$tag.json
done
PASSES=kt PASS_TIMEOUT=240 bash tools/prof_step.sh cfg2_actor_fp32 --workload cfg2 --actor mlp_fp32 > $O/r06_cfg2_actor_fp32_rocprofv3.txt 2>&1; tail -6 $O/r06_cfg2_actor_fp32_rocprofv3.txt
PASSES=kt PASS_TIMEOUT=240 bash tools/prof_step.sh cfg2_actor_bf16 --workload cfg2 --actor mlp > $O/r06_cfg2_actor_bf16_rocprofv3.txt 2>&1; tail -6 $O/r06_cfg2_actor_bf16_rocprofv3.txt
python tools/collect_evidence.py $O/r06_hbm_traffic.json cfg2_persistent=cfg2:persistent cfg2_per_step=cfg2:per_step cfg3_persistent=cfg3:persistent cfg4_persistent=cfg4:persistent > /dev/null
rm -rf gpurun_out/prof_*   # raw rocprofv3 output: too large to travel back (the summaries above carry what is committed)
timeout 200 python tools/refill_time.py cfg2 2>&1 | tail -1 > $O/r06_refill_time.txt; timeout 200 python tools/refill_time.py cfg3 2>&1 | tail -1 >> $O/r06_refill_time.txt; cut -c1-160 $O/r06_refill_time.txt
timeout 200 python tools/sb3_collect_bench.py cfg2 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r06_collector_cfg2.json; timeout 200 python tools/sb3_collect_bench.py cfg3 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r06_collector_cfg3.json; timeout 200 python tools/sb3_collect_bench.py cfg2 12 fp32 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r06_collector_cfg2_fp32.json; cut -c1-200 $O/r06_collector_cfg3.json
for w in cfg2 cfg3 cfg4; do timeout 200 python tools/stats_time.py $w 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/r06_stats_time.txt
EV2G_PT_LIB=build_variants/libev2g_pt.so timeout 300 python tools/phase_timing.py cfg4 2>&1 | grep -v amdgpu.ids | head -9 > $O/r06_phase_cfg4.txt
timeout 200 python tools/sb3_loop_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r06_host_loops.txt; SB3_COPY_OBS=0 timeout 200 python tools/sb3_loop_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/^/copy_obs=False: /' >> $O/r06_host_loops.txt
timeout 200 python tools/facade_loop_bench.py 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/r06_host_loops.txt; cat $O/r06_host_loops.txt
