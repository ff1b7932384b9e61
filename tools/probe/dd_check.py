"""exact check (fractions.Fraction; the square root through 60-digit mpmath) of the device results written by tools/probe/dd_check"""
import sys
from fractions import Fraction as F
import numpy as np
import mpmath as mp
mp.mp.dps = 60
n = 1 << 16
raw = np.fromfile("gpurun_out/dd_check.bin", dtype=np.float64)
ah, al, bh, bl, d = (raw[i * n:(i + 1) * n] for i in range(5))
out = raw[5 * n:].reshape(14, n)
names = ["add", "mul", "mul_d", "rsqrt", "sub", "two_prod", "rsqrt_1"]
worst = [0.0] * 7
for i in range(0, n, 16):
    A, B = F(ah[i]) + F(al[i]), F(bh[i]) + F(bl[i])
    for q in range(7):
        got = F(out[2 * q, i]) + F(out[2 * q + 1, i])
        if q == 0: want, scale = A + B, abs(A) + abs(B)
        elif q == 1: want = A * B; scale = abs(want)
        elif q == 2: want = A * F(d[i]); scale = abs(want)
        elif q in (3, 6):
            X = F(abs(ah[i]) + 0.1) + F(al[i])
            w = 1 / mp.sqrt(mp.mpf(X.numerator) / mp.mpf(X.denominator))
            e = abs((mp.mpf(got.numerator) / mp.mpf(got.denominator) - w) / w)
            worst[q] = max(worst[q], float(e)); continue
        elif q == 4: want, scale = A - B, abs(A) + abs(B)
        else: want = F(ah[i]) * F(bh[i]); scale = abs(want)
        if scale != 0:
            worst[q] = max(worst[q], float(abs(got - want) / scale))
for nm, w in zip(names, worst):
    print("%-8s worst relative error on the device %.2e" % (nm, w))
