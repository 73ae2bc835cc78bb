#!/bin/bash
# round 5, call A: parity suite on the dictionary / strided kernels, A/B against round 4's library, default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for w in cfg2 cfg3; do timeout 600 python tools/ab_bench.py --workload $w --reps 30 --pool 4 build_variants/r5_base.so ev2gym_amd/libev2g_hip.so ev2gym_amd/libev2g_hip.so@EV2G_NO_DICT=1 2>&1 | grep -v amdgpu.ids | sed 's/   digest \[.*//' | tee $O/ab_$w.txt; done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -c 300 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5a/bench_default.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "launch_us", d["roofline"]["avg_launch_us"])
print("per-step frac", d["roofline_by_launch_mode"]["per_step"]["frac"], d["roofline_by_launch_mode"]["per_step"]["avg_launch_us"])
print("full_episode", d["full_episode"])
print("strided", json.dumps(d.get("persistent_strided")))
print("others", json.dumps(d.get("other_workloads")))
print("rollout", {k: v for k, v in d["rollout"].items() if k != "collector" and k != "note"})
print("refill", {k: v for k, v in d["device_refill"].items() if k != "note"})
PY
