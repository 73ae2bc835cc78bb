cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4ev3; mkdir -p $O
timeout 300 python bench.py --workload cfg3 > $O/r04_bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
bash tools/prof_step.sh cfg3_persistent --workload cfg3 --launch persistent > $O/r04_cfg3_persistent_rocprofv3.txt 2>&1; tail -8 $O/r04_cfg3_persistent_rocprofv3.txt
mkdir -p $O/summaries; cp gpurun_out/prof_cfg3_persistent/summary.json $O/summaries/cfg3_persistent.json; rm -rf gpurun_out/prof_*
