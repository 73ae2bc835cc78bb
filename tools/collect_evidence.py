#!/usr/bin/env python3
"""After tools/prof_step.sh runs: gather the per-launch HBM counters of the step kernels into one JSON (run on the GPU box).
usage: python tools/collect_evidence.py <out.json> tag=workload:launch ...   (tag = the <tag> given to prof_step.sh)"""
import json, sys
out, specs = sys.argv[1], sys.argv[2:]
res = {"_comment": "HBM bytes per step-kernel launch from rocprofv3 PMC passes (tools/prof_step.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc "
                   "runs, KB, average over the dispatches of the step kernel); bench.py reports 2*FETCH + WRITE (gfx950 FETCH_SIZE note in "
                   "MI355X_MICROARCH.md).  Every entry is the summary.json of one profiles/rNN_<tag>_rocprofv3.txt"}
for spec in specs:
    tag, wl = spec.split("=")
    workload, launch = wl.split(":")
    try:
        e = json.load(open(f"gpurun_out/prof_{tag}/summary.json"))
    except Exception as ex:
        print("skip", tag, ex)
        continue
    if "fetch_kb" in e:
        res.setdefault(workload, {})[launch] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
