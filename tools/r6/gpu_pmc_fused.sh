#!/bin/bash
# PMC passes over the fused actor + step launches (separate rocprofv3 runs, --pmc only): MFMA busy, instruction mix, waits, LDS conflicts, L2 hits -- bf16 and float32 policies
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r6_pmc_fused; mkdir -p $O; rm -rf $O/p*
for A in mlp mlp_fp32; do
CMD="python bench.py --no-cpu-baseline --only-timed --min-time 0.1 --pool 2 --workload cfg2 --actor $A"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $line -d $O/${A}_p$i -o p$i -- $CMD > $O/${A}_p$i.log 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
PASSES
done
python - "$O" <<'PY' | tee $O/pmc.txt
import sqlite3, sys, os, glob
o = sys.argv[1]
for d in sorted(glob.glob(o + "/mlp*_p*/")):
    n = os.path.basename(d.rstrip("/"))
    db = glob.glob(d + "/*_results.db")
    if not db:
        print(f"PMC | {n}: no database; log tail:", open(f"{o}/{n}.log").read()[-200:].replace("\n", " | "))
        continue
    con = sqlite3.connect(f"file:{db[0]}?mode=ro", uri=True)
    for r in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%ev2g_step_wave%' group by kernel_name, counter_name"):
        print("PMC |", n, "|", r[0][5:52], "|", r[1], "|", round(r[2], 1), "| n =", r[3])
PY
