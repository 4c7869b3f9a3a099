"""What the two halves of one step of the decoder's dropout generator are worth as Bernoulli(p) decisions (csrc/dec_w.hip: decw_step).
x <- x[23:0] * 214013 + c; the upper half of the 32-bit result is the middle of the product, the lower half is by itself the full-period
16-bit generator x[15:0] * 17405 + c.  Over the full 2^24 period, per increment: single rates, the joint rate of a step's two decisions, of
successive decisions (lag 1, lag 2), of three successive ones, and the histogram of drops among 12 successive draws of the lower half against
the binomial.  No GPU.      python tools/checks/lcg_halves.py [p]"""
import math
import sys
import numpy as np

A = 214013
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
t = int(p * 65536)
for c in (2531011, 1013904223 & 0xffffff, 12345, 7046029):
    x = np.arange(1 << 24, dtype=np.uint64)
    steps, cur = [], x
    for _ in range(12):
        r = (cur * A + c) & 0xffffffff
        steps.append(((r >> 16) < t, (r & 0xffff) < t))
        cur = r & 0xffffff
    (h0, l0), (h1, l1), (h2, l2) = steps[:3]
    print(f"c = {c}: rate upper {h0.mean():.6f} lower {l0.mean():.6f} (p = {t / 65536:.6f}); same step upper & lower {(h0 & l0).mean():.6f} (p^2 = {(t / 65536) ** 2:.6f})")
    print(f"   lag 1: upper {(h0 & h1).mean():.6f} lower {(l0 & l1).mean():.6f} upper-lower {(h0 & l1).mean():.6f} lower-upper {(l0 & h1).mean():.6f};  lag 2: upper {(h0 & h2).mean():.6f} lower {(l0 & l2).mean():.6f}")
    print(f"   three successive: upper {(h0 & h1 & h2).mean():.6f} lower {(l0 & l1 & l2).mean():.6f} (p^3 = {(t / 65536) ** 3:.6f})")
    cnt = np.sum([s[1] for s in steps], axis=0)
    hist = np.bincount(cnt, minlength=13) / cnt.size
    q = t / 65536
    binom = [math.comb(12, k) * q ** k * (1 - q) ** (12 - k) for k in range(13)]
    print("   drops among 12 successive lower halves:", np.round(hist[:6], 5), "binomial", np.round(binom[:6], 5))
