#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/prof_refill; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o rf -- python tools/refill_time.py ${1:-cfg2} > $O/run.log 2>&1
tail -2 $O/run.log
python - <<P
import sqlite3, glob
db = glob.glob("$O/**/rf_results.db", recursive=True)
con = sqlite3.connect(db[0])
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"): print("KT |", r[0][:70], "|", r[1], "|", round(r[2],1), "|", round(r[3],2), "|", round(r[4],1))
P
