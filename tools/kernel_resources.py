#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the HIP library (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ev2gym_amd", "csrc", "ev2g_host.hip")


def main():
    extra = sys.argv[1:]
    sys.path.insert(0, ROOT)
    from ev2gym_amd import build
    cmd = [build.hipcc()] + build.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_ev2g_res.so", SRC] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: .*?:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
        if "error" in line:
            print(line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':60s} VGPR AGPR SGPR sSpill vSpill scratch occ  LDS")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{name:60s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('SGPRs','?'):>4s} {r.get('SGPRs Spill','?'):>6s} "
              f"{r.get('VGPRs Spill','?'):>6s} {r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>3s} {r.get('LDS Size [bytes/block]','?'):>5s}")


if __name__ == "__main__":
    main()
