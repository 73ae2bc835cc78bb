"""Builds libev2g_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No CPU fallback exists."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libev2g_hip.so")
SRC = os.path.join(HERE, "csrc", "ev2g_host.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("ev2g_device.h", "ev2g_step_v2.h", "ev2g_step_wave.h", "ev2g_step_big.h", "ev2g_mlp.h", "ev2g_comm.h", "ev2g_gen.h", "ev2g_gen_host.h", "ev2g_refill.h", "ev2g_refill_host.h")] + [
    os.path.join(HERE, "..", "include", "ev2g.h")]
# -ffp-contract=off: the reference's operation order must survive (EV.my_ceil, ev.py:188-189)
# -disable-machine-licm: the step kernels sit at the 128-VGPR / 4-waves-per-SIMD boundary; machine LICM hoists the constant
#   materialisations of the inlined float64 exp / division sequences out of the step loop and the register allocator then
#   spills them to scratch and reloads them inside the dependent chain of the battery maths (measured: DESIGN.md §3)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-mllvm", "-disable-machine-licm"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the EV2Gym step engine is HIP-only (gfx950)")


def extra_drop(extra):
    return [f[len("--drop="):] for f in extra if f.startswith("--drop=")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    m = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > m for d in DEPS)


def build(force=False, verbose=False, out=None, extra=()):
    """Build the library in-tree; `out` / `extra` produce A/B variants (tools/ab_bench.py) next to it."""
    if out is not None:
        cmd = [hipcc()] + [f for f in FLAGS if f not in extra_drop(extra)] + [f for f in extra if not f.startswith("--drop=")] + ["-o", out, SRC]
        subprocess.check_call(cmd)
        return out
    if force or needs_build():
        cmd = [hipcc()] + FLAGS + ["-o", LIB, SRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
