#!/bin/bash
# PMC passes over the cfg4 step kernel (separate rocprofv3 runs, --pmc only): bash tools/r6/gpu_pmc.sh <out> [env VAR=..]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/$1; shift; mkdir -p $O; rm -rf $O/p*
CMD="python tools/r6/run_cfg.py ${WL:-cfg4} 3"
$CMD 2>&1 | grep -v amdgpu.ids | tee $O/run.txt
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d $O/p$i -o p$i -- $CMD > $O/p$i.log 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_SMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum
GRBM_GUI_ACTIVE TCC_BUSY_avr TCC_TAG_STALL_sum
FETCH_SIZE
WRITE_SIZE
PASSES
python - "$O" <<'PY'
import sqlite3, sys, os, glob
o = sys.argv[1]
for d in sorted(glob.glob(o + "/p*/"), key=lambda s: int(s.rstrip("/").split("/p")[-1])):
    n = os.path.basename(d.rstrip("/"))
    db = glob.glob(d + "/*_results.db")
    if not db:
        print(f"PMC | {n}: no database; log tail:", open(f"{o}/{n}.log").read()[-300:].replace("\n", " | "))
        continue
    con = sqlite3.connect(f"file:{db[0]}?mode=ro", uri=True)
    for r in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%ev2g_step_%' group by kernel_name, counter_name"):
        print("PMC |", r[0][:30], "|", r[1], "|", round(r[2], 1), "| n =", r[3])
PY
