#!/usr/bin/env python3
"""Static instruction mix per phase of one step kernel.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DEV2G_PHASE_MARKERS -S --cuda-device-only \
        -o /tmp/ev2g_markers.s ev2gym_amd/csrc/ev2g_host.hip
  python tools/isa_phase_count.py /tmp/ev2g_markers.s _Z14ev2g_step_waveILi0ELi0EEvPK3V2P6StepIOiii

PT_MARK(i) closes the region that is accounted to phase i (same convention as tools/phase_timing.py).  Counts are
static (every instruction once, whatever the branch structure), so they bound the dynamic per-wave mix from above.
"""
import collections
import re
import sys

path, fn = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(fn + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
names = {0: "A", 1: "barrier", 2: "B", 3: "C", 4: "D", 5: "E", 6: "prefetch", 7: "loop-top/prologue"}
cur = collections.Counter()
tot = collections.defaultdict(collections.Counter)
order = []
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; PHASE_MARK (\d+)", t)
    if m:
        k = int(m.group(1))
        tot[k].update(cur)
        order.append(k)
        cur = collections.Counter()
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    cls = ("valu_f64" if re.match(r"v_.*_f64", op) else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_barrier", "s_load", "s_cbranch", "s_branch", "s_nop"))
           else "smem" if op.startswith("s_load") else "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith(("s_waitcnt", "s_barrier", "s_nop"))
           else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cur[cls] += 1
    if op in ("v_rcp_f64_e32", "v_exp_f32_e32", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_rcp_f64_e64"):
        cur["(" + op + ")"] += 1
tot["epilogue"].update(cur)
cols = ["valu", "valu_f64", "salu", "smem", "branch", "wait", "lds", "vmem", "(v_div_fixup_f64)", "(v_rcp_f64_e32)"]
print(f"{'phase':20s}" + "".join(f"{c:>18s}" for c in cols))
for k in [7, 0, 6, 1, 2, 3, 4, 5, "epilogue"]:
    if k in tot:
        print(f"{names.get(k, k):20s}" + "".join(f"{tot[k][c]:18d}" for c in cols))
