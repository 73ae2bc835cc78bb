#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box via gpurun): kernel trace + PMC in separate runs.
# PASSES=kt: kernel trace only.  usage: bash tools/prof_step.sh <tag> <bench.py args...> ; text summary on stdout, raw output under gpurun_out/prof_<tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --min-time 0.1 $*"
echo "## command: rocprofv3 <mode> -- $B"
$B 2>/dev/null | tail -1 > $OUT/bench_line.json; cat $OUT/bench_line.json
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
if [ "${PASSES:-all}" = all ]; then
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $OUT/pmc2 -o pmc2 -- $B > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $B > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc4 -o pmc4 -- $B > $OUT/pmc4.log 2>&1
fi
python - $OUT <<'P'
import sqlite3, sys
out = sys.argv[1]
con=sqlite3.connect(f'{out}/kt/kt_results.db')
print("## kernel trace (--kernel-trace --stats): name | calls | total us | avg us | %")
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"): print("KT |", r[0][:70], "|", r[1], "|", round(r[2],1), "|", round(r[3],3), "|", round(r[4],2))
print("## PMC, average per dispatch of the step kernel")
for d in ['pmc1','pmc2','pmc3','pmc4']:
    try:
        con=sqlite3.connect(f'{out}/{d}/{d}_results.db')
        for r in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%ev2g_step_%' group by kernel_name, counter_name"): print("PMC |", r[0][:34], "|", r[1], "|", round(r[2],1), "| n =", r[3])
    except Exception as e: print(d,'ERR',e)
P
