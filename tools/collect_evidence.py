#!/usr/bin/env python3
"""After tools/prof_step.sh runs: gather the per-launch HBM counters of the step kernels into one JSON (run on the GPU box).
usage: python tools/collect_evidence.py <out.json> tag=workload:launch ...   (tag = the <tag> given to prof_step.sh)"""
import json, sqlite3, sys
out, specs = sys.argv[1], sys.argv[2:]
res = {"_comment": "HBM bytes per step-kernel launch from rocprofv3 PMC passes (tools/prof_step.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc "
                   "runs, KB, average over the dispatches of the step kernel); bench.py reports 2*FETCH + WRITE (gfx950 FETCH_SIZE note in "
                   "MI355X_MICROARCH.md)"}
for spec in specs:
    tag, wl = spec.split("=")
    workload, launch = wl.split(":")
    e = {}
    for d, ctr, key in (("pmc3", "FETCH_SIZE", "fetch_kb"), ("pmc4", "WRITE_SIZE", "write_kb")):
        con = sqlite3.connect(f"gpurun_out/prof_{tag}/{d}/{d}_results.db")
        r = con.execute("select kernel_name, avg(value), count(*) from counters_collection where kernel_name like '%ev2g_step_%' and counter_name=? "
                        "group by kernel_name order by count(*) desc", (ctr,)).fetchone()
        e["kernel"], e[key], e["dispatches"] = r[0].split("(")[0].replace("void ", ""), round(r[1], 1), r[2]
    bl = json.loads(open(f"gpurun_out/prof_{tag}/bench_line.json").read())
    e["steps_per_launch"] = bl["roofline_by_launch_mode"][launch]["steps_per_launch"]
    res.setdefault(workload, {})[launch] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
