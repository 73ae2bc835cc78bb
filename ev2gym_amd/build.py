"""Builds libev2g_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No CPU fallback exists."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libev2g_hip.so")
SRC = os.path.join(HERE, "csrc", "ev2g_host.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("ev2g_device.h", "ev2g_step_v2.h", "ev2g_step_wave.h")] + [
    os.path.join(HERE, "..", "include", "ev2g.h")]
# -ffp-contract=off: the reference's operation order must survive (EV.my_ceil, ev.py:188-189)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the EV2Gym step engine is HIP-only (gfx950)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    m = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > m for d in DEPS)


def build(force=False, verbose=False):
    if force or needs_build():
        cmd = [hipcc()] + FLAGS + ["-o", LIB, SRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
