#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box via gpurun): kernel trace + PMC in separate runs (gpurun refuses --pmc combined
# with the hip/hsa trace domains; --kernel-trace/--stats only).  The text summary on stdout is what gets committed under profiles/:
# it carries the bench line, the kernel trace, EVERY PMC block (or an explicit "not collected"), and the figures recomputed from them.
# usage: bash tools/prof_step.sh <tag> <bench.py args...>     raw output under gpurun_out/prof_<tag>
#   every rocprofv3 pass runs under `timeout` (PASS_TIMEOUT, default 300 s): a pass that hangs costs its limit, not the call's
#   PASSES=kt   kernel trace only (the summary then says so instead of printing counter blocks)
#   bench.py runs with --only-timed: nothing but the timed regions of the chosen launch mode, so a `--launch per_step` trace holds
#   single-step dispatches only and a `--launch persistent` trace whole-episode dispatches only.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --only-timed --min-time ${MIN_TIME:-0.1} --pool ${POOL:-2} $*"
echo "## command: rocprofv3 <mode> -- $B"
$B 2>$OUT/bench.err | tail -1 > $OUT/bench_line.json
timeout ${PASS_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
if [ "${PASSES:-all}" = all ]; then
timeout ${PASS_TIMEOUT:-300} rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- $B > $OUT/pmc1.log 2>&1
timeout ${PASS_TIMEOUT:-300} rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $OUT/pmc2 -o pmc2 -- $B > $OUT/pmc2.log 2>&1
timeout ${PASS_TIMEOUT:-300} rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $B > $OUT/pmc3.log 2>&1
timeout ${PASS_TIMEOUT:-300} rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc4 -o pmc4 -- $B > $OUT/pmc4.log 2>&1
fi
python tools/prof_summary.py $OUT "${PASSES:-all}"
