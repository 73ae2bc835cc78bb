#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box via gpurun): kernel trace + PMC in separate runs.
# usage: bash tools/prof_step.sh [workload] [launch] ; summaries are printed, raw output under gpurun_out/prof
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
W=${1:-cfg2}; L=${2:-per_step}
B="python bench.py --workload $W --steps 224 --warmup 112 --launch $L --no-cpu-baseline"
rm -rf gpurun_out/prof/*
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/kt -o kt -- $B > gpurun_out/prof/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/prof/pmc1 -o pmc1 -- $B > gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES -d gpurun_out/prof/pmc2 -o pmc2 -- $B > gpurun_out/prof/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d gpurun_out/prof/pmc3 -o pmc3 -- $B > gpurun_out/prof/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/prof/pmc4 -o pmc4 -- $B > gpurun_out/prof/pmc4.log 2>&1
python - <<'P'
import sqlite3,glob
con=sqlite3.connect('gpurun_out/prof/kt/kt_results.db')
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 6"): print("KT", r)
for d in ['pmc1','pmc2','pmc3','pmc4']:
    try:
        con=sqlite3.connect(f'gpurun_out/prof/{d}/{d}_results.db')
        for r in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%step%' group by kernel_name, counter_name"): print(d, r[0][:24], r[1], round(r[2],1), r[3])
    except Exception as e: print(d,'ERR',e)
P
