#!/usr/bin/env python3
"""Text + JSON summary of one tools/prof_step.sh run (on the GPU box): bench line, kernel trace, every PMC block, and the
figures derived from them -- so that profiles/<round>_<tag>_rocprofv3.txt can be checked without anything else.
usage: python tools/prof_summary.py gpurun_out/prof_<tag> [all|kt]"""
import json
import os
import sqlite3
import sys

out, passes = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "all")
HBM_PEAK = 8000.0
summ = {"tag": os.path.basename(out).replace("prof_", "")}
try:   # the commit this tree was pushed from (written by the caller before gpurun: the GPU box has no .git)
    summ["source"] = open(".head_sha").read().strip()
    print("## source:", summ["source"])
except Exception:
    pass

try:
    line = json.loads(open(f"{out}/bench_line.json").read())
    print(json.dumps(line))
except Exception as e:   # the un-profiled run failed: say why
    line = None
    print("## bench line missing:", e)
    print(open(f"{out}/bench.err").read()[-2000:])

step_avg_us = None
try:
    con = sqlite3.connect(f"file:{out}/kt/kt_results.db?mode=ro", uri=True)
    print("## kernel trace (--kernel-trace --stats): name | calls | total us | avg us | %")
    for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 10"):
        print("KT |", r[0][:70], "|", r[1], "|", round(r[2], 1), "|", round(r[3], 3), "|", round(r[4], 2))
        if "ev2g_step_" in r[0] and step_avg_us is None:
            step_avg_us, summ["kernel"], summ["dispatches"] = r[3], r[0].split("(")[0].replace("void ", ""), r[1]
except Exception as e:
    print("## kernel trace: NOT AVAILABLE:", e)
    try:
        print(open(f"{out}/kt.log").read()[-1500:])
    except Exception:
        pass

pmc = {}
if passes != "all":
    print(f"## PMC: not collected in this run (PASSES={passes}: kernel trace only)")
else:
    print("## PMC, average per dispatch of the step kernel (separate rocprofv3 --pmc runs)")
    for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
        path = f"{out}/{d}/{d}_results.db"
        if not os.path.exists(path):
            print(f"PMC | {d}: pass produced no database; tail of its log:")
            try:
                print(open(f"{out}/{d}.log").read()[-800:])
            except Exception:
                pass
            continue
        con = sqlite3.connect(f"file:{path}?mode=ro", uri=True)
        for r in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                             "where kernel_name like '%ev2g_step_%' group by kernel_name, counter_name"):
            print("PMC |", r[0][:34], "|", r[1], "|", round(r[2], 1), "| n =", r[3])
            pmc[r[1]] = r[2]

if line and step_avg_us:
    cfg = line["config"]
    spl = cfg["steps_per_episode"] if cfg["launch"] == "persistent" else 1
    if "1024" in summ.get("kernel", "") and "true" in summ.get("kernel", "").split("1024")[-1]:   # the fused actor + step launch: one launch per episode-long segment
        spl = cfg["steps_per_episode"]
        print("## the step kernel of this trace is the FUSED actor + step launch (the policy inside it): bytes below are the env step's algorithmic bytes only")
    b = cfg["algorithmic_bytes_per_env_step"] * cfg["envs_per_gpu"] * spl
    ach = b / (step_avg_us * 1e-6) / 1e9
    summ.update(launch=cfg["launch"], steps_per_launch=spl, avg_launch_us=step_avg_us, algorithmic_bytes_per_launch=b,
                achieved_GBps=ach, frac=ach / HBM_PEAK)
    print("## derived")
    print(f"DER | step kernel {summ['kernel']}: {step_avg_us:.3f} us per launch of {spl} step(s) = {step_avg_us / spl:.3f} us/step")
    print(f"DER | algorithmic bytes per launch = {cfg['algorithmic_bytes_per_env_step']:.1f} B x {cfg['envs_per_gpu']} envs x {spl} = {b / 1e6:.2f} MB"
          f" -> {ach:.1f} GB/s = {ach / HBM_PEAK:.4f} of {HBM_PEAK:.0f} GB/s")
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        tr = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        summ.update(fetch_kb=pmc["FETCH_SIZE"], write_kb=pmc["WRITE_SIZE"], traffic_bytes=tr, traffic_over_algorithmic=tr / b)
        print(f"DER | HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 FETCH_SIZE note, MI355X_MICROARCH.md) = {tr / 1e6:.2f} MB"
              f" = {tr / b:.2f} x algorithmic")
    if "SQ_WAIT_ANY" in pmc and "SQ_WAVE_CYCLES" in pmc:
        summ["wait_any_over_wave_cycles"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
        print(f"DER | SQ_WAIT_ANY / SQ_WAVE_CYCLES = {summ['wait_any_over_wave_cycles']:.3f}")
    if "SQ_INSTS_SALU" in pmc and "SQ_INSTS_VALU" in pmc:
        summ["salu_over_valu"] = pmc["SQ_INSTS_SALU"] / pmc["SQ_INSTS_VALU"]
        print(f"DER | SALU : VALU = {summ['salu_over_valu']:.3f}")
    if "SQ_LDS_BANK_CONFLICT" in pmc and pmc.get("SQ_LDS_IDX_ACTIVE"):
        summ["lds_conflict_ratio"] = pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_LDS_IDX_ACTIVE"]
        print(f"DER | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {summ['lds_conflict_ratio']:.3f}")
    if "SQ_ACTIVE_INST_VALU" in pmc and "SQ_WAVE_CYCLES" in pmc:
        print(f"DER | SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {pmc['SQ_ACTIVE_INST_VALU'] / pmc['SQ_WAVE_CYCLES']:.3f}")
json.dump(summ, open(f"{out}/summary.json", "w"), indent=1)
