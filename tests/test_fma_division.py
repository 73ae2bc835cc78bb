"""The two constant divisions the step kernels replace by `q0 = n*RN(1/b); q = fma(fma(-q0, b, n), RN(1/b), q0)`
(ev2gym_amd/csrc/ev2g_device.h: div_int_by_const) are bit-identical to the IEEE division on the WHOLE range the
device code uses the short form for -- checked exhaustively here with the host's correctly rounded fma():
  * round(a, 5)  (ev_charger.py:157):  n / 1e5 for every integer |n| <= 2e5
  * my_ceil      (ev.py:188-189):      n / 100 for every integer |n| <= 2e7
"""
import os
import subprocess
import tempfile

SRC = r"""
#include <math.h>
#include <stdio.h>
static double qdiv(double n, double b, double y) { double q0 = n * y; double r = fma(-q0, b, n); return fma(r, y, q0); }
int main(void) {
    long bad5 = 0, bad2 = 0;
    volatile double b5 = 100000.0, b2 = 100.0;
    const double y5 = 1.0 / b5, y2 = 1.0 / b2;
    for (long n = -200000; n <= 200000; n++) if (qdiv((double)n, b5, y5) != (double)n / b5) bad5++;
    for (long n = -20000000; n <= 20000000; n++) if (qdiv((double)n, b2, y2) != (double)n / b2) bad2++;
    printf("%ld %ld\n", bad5, bad2);
    return 0;
}
"""


def test_short_division_sequences_are_exact_on_their_whole_range():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "chk.c"), os.path.join(d, "chk")
        open(src, "w").write(SRC)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"])
        out = subprocess.check_output([exe]).decode().split()
    assert out == ["0", "0"], out
