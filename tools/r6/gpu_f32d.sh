#!/bin/bash
# round 6, last session: the adopted float32 fused launch (group 2, ring 4) -- parity tests, the fused regressions, the default bench line with rollout.fp32
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f32d; mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "float32" 2>&1 | tail -5 | tee $O/pytest_new.txt
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_actor_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_regress.txt
timeout 600 python bench.py 2>$O/err.txt | tail -1 > $O/bench_default.json
python -c "
import json
d=json.loads(open('$O/bench_default.json').read())
print('value', d['value'], 'frac', d['roofline']['frac'])
r=d['rollout']; print('rollout bf16', r['env_steps_per_s_per_gpu'], r['fused_launch'], 'collector', r.get('collector',{}).get('env_steps_per_s'))
print('rollout fp32', json.dumps(r.get('fp32')))
print('refill', d['device_refill']['ms_per_episode_with_refill'], 'full_episode', d['full_episode']['ms_per_episode'])
" | tee $O/summary.txt
tail -3 $O/err.txt
