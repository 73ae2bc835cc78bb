#!/bin/bash
# PMC passes over ev2g_refill_kernel (separate rocprofv3 runs, --pmc only): is the refill issue-bound, latency-bound or write-bound?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r6_pmc_refill; mkdir -p $O; rm -rf $O/p*
CMD="python tools/refill_time.py ${WL:-cfg2}"
$CMD 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee $O/run.txt
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d $O/p$i -o p$i -- $CMD > $O/p$i.log 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_SMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_IFETCH SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum
GRBM_GUI_ACTIVE TCC_BUSY_avr TCC_TAG_STALL_sum
FETCH_SIZE
WRITE_SIZE
PASSES
python - "$O" <<'PY' | tee $O/pmc.txt
import sqlite3, sys, os, glob
o = sys.argv[1]
for d in sorted(glob.glob(o + "/p*/"), key=lambda s: int(s.rstrip("/").split("/p")[-1])):
    n = os.path.basename(d.rstrip("/"))
    db = glob.glob(d + "/*_results.db")
    if not db:
        print(f"PMC | {n}: no database; log tail:", open(f"{o}/{n}.log").read()[-300:].replace("\n", " | "))
        continue
    con = sqlite3.connect(f"file:{db[0]}?mode=ro", uri=True)
    for r in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%refill%' group by kernel_name, counter_name"):
        print("PMC |", r[0][:30], "|", r[1], "|", round(r[2], 1), "| n =", r[3])
PY
