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


# ---- round 4: divisions by per-session and physical constants through correctly rounded reciprocals (ev2g_fdiv1 / ev2g_fdiv2) ----
def _delta_in_units_of_2_pow_minus_54(b):
    """|b * RN(1/b) - 1| / 2^-54, in exact rational arithmetic."""
    from fractions import Fraction
    return abs(Fraction(b) * Fraction(1.0 / b) - 1) * 2 ** 54


def test_constants_divided_with_one_correction_step_have_a_small_reciprocal_error():
    """ev2g_fdiv1 is exact for EVERY numerator when |b*RN(1/b) - 1| <= 2^-54 (then RN(a*RN(1/b)) is a faithful quotient and Markstein's
    theorem applies; ev2g_device.h).  The kernels use it for 1000, 60 and -- when the host's check at load passes -- the step length."""
    for b in (1000.0, 60.0, 100.0, 15.0, 30.0, 60.0, 5.0, 10.0, 20.0):
        assert _delta_in_units_of_2_pow_minus_54(b) <= 1, b
    assert _delta_in_units_of_2_pow_minus_54(100000.0) > 1    # ... which is why round(a, 5) keeps its exhaustively checked range
    assert _delta_in_units_of_2_pow_minus_54(49.0) > 1        # a step length of 49 minutes would take the division (V2P::dt_fdiv == 0); every length up to 48 qualifies
    assert all(_delta_in_units_of_2_pow_minus_54(float(b)) <= 1 for b in range(1, 49))


SRC2 = r"""
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
static uint64_t s = 88172645463325252ULL;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double mk(uint64_t m, int e) { union { uint64_t u; double d; } x; x.u = ((uint64_t)(1023 + e) << 52) | (m & 0xFFFFFFFFFFFFFULL); return x.d; }
static double fdiv1(double a, double b, double rb) { double q0 = a * rb; return fma(fma(-q0, b, a), rb, q0); }
static double fdiv2(double a, double b, double rb) { double q1 = fdiv1(a, b, rb); return fma(fma(-q1, b, a), rb, q1); }
int main(int argc, char **argv) {
    long n = atol(argv[1]), bad1 = 0, bad2 = 0, badc = 0;
    const double consts[5] = {1000.0, 60.0, 15.0, 30.0, 100.0};
    for (long i = 0; i < n; i++) {
        volatile double b = mk(rnd(), (int)(rnd() % 24) - 12), a = mk(rnd(), (int)(rnd() % 24) - 12);
        if (i % 3 == 1) { a = mk(rnd(), 0) * b; }                      /* quotients next to representable numbers */
        if (i % 3 == 2) { double q = mk(rnd() | 1, 0); a = (q + ((rnd() & 1) ? 0x1p-53 : -0x1p-53)) * b; }   /* ... and next to rounding midpoints */
        if (rnd() & 1) a = -a;
        const double rb = 1.0 / b, t = a / b;
        if (fdiv1(a, b, rb) != t) bad1++;
        if (fdiv2(a, b, rb) != t) bad2++;
        volatile double c = consts[i % 5];
        if (fdiv1(a, c, 1.0 / c) != a / c) badc++;
    }
    printf("%ld %ld %ld\n", bad1, bad2, badc);
    return 0;
}
"""


def test_reciprocal_divisions_equal_the_ieee_division_on_random_and_adversarial_operands():
    """2e8 operand pairs (random, quotients next to floating-point numbers, quotients next to rounding midpoints): the two-step form
    (per-session divisors B and v) and the one-step form on the qualified constants never differ from the IEEE division."""
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "chk2.c"), os.path.join(d, "chk2")
        open(src, "w").write(SRC2)
        subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, src, "-lm"])
        bad1, bad2, badc = (int(x) for x in subprocess.check_output([exe, "200000000"]).decode().split())
    assert bad2 == 0 and badc == 0, (bad1, bad2, badc)
