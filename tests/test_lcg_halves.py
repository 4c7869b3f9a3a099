"""The decoder's dropout generator (csrc/dec_w.hip: decw_step) yields two 16-bit uniforms per step -- the upper half of x * A + c and the
lower half, which by itself is the 16-bit generator x[15:0] * (A mod 2^16) + c.  This restates the step in numpy and holds the Bernoulli(0.1)
decisions of both halves to the first- and second-order statistics of independent draws over the generator's full period (the kernel-side
rates are tests/test_gpu_dropstats.py's; tools/checks/lcg_halves.py prints the long table).  No GPU."""
import numpy as np

A, C, T = 214013, 2531011, int(0.1 * 65536)


def _step(x):
    r = (x * A + C) & 0xffffffff
    return r & 0xffffff, (r >> 16) < T, (r & 0xffff) < T


def test_both_halves_of_a_step_are_bernoulli_decisions_with_independent_pairs():
    x = np.arange(1 << 24, dtype=np.uint64)
    x1, h0, l0 = _step(x)
    _, h1, l1 = _step(x1)
    p = T / 65536
    assert abs(h0.mean() - p) < 2e-6 and abs(l0.mean() - p) < 2e-6                    # full period: every 16-bit value of the lower half equally often
    for name, both in (("upper & lower of one step", h0 & l0), ("upper, next upper", h0 & h1), ("lower, next lower", l0 & l1),
                       ("upper, next lower", h0 & l1), ("lower, next upper", l0 & h1)):
        assert abs(both.mean() / (p * p) - 1.0) < 5e-3, (name, both.mean())         # measured: within 0.3 % of p^2
    assert (A % 65536) % 4 == 1 and C % 2 == 1                                       # full period 2^16 of the lower half (Hull-Dobell)
