#!/usr/bin/env python3
"""
Table-assisted 2^t for the dense kernel (abd_dense.hpp):  2^t = 2^e * T[j] * P(f),  1024 t = 1024 e + j + f,
|f| <= 1/2, T[j] = 2^(j/1024) correctly rounded, P(f) = 1 + f (c1 + f (c2 + f c3)) ~ 2^(f/1024).

Prints the near-minimax coefficients (Chebyshev interpolation of (2^(f/1024) - 1) / f in 60-digit arithmetic) and
the measured worst relative error of the whole scheme evaluated in IEEE double (fma emulated exactly with
fractions) against 60-digit references.
"""
import sys
from decimal import Decimal as D, getcontext
from fractions import Fraction as F
import math
import random

getcontext().prec = 60
LN2 = D(2).ln()
NTAB = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
DEG = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def cos_dec(x):
    x = D(x)
    s, term, k = D(0), D(1), 0
    while abs(term) > D(10) ** -58:
        s += term
        k += 2
        term = -term * x * x / (k * (k - 1))
    return s


PI = D("3.14159265358979323846264338327950288419716939937510582097494")


def g(f):  # (2^(f/NTAB) - 1) / f
    if abs(f) < D(10) ** -25:
        return LN2 / NTAB
    return ((f * LN2 / NTAB).exp() - 1) / f


n = DEG  # g is approximated by degree DEG-1
h = D("0.5")
nodes = [h * cos_dec(PI * (2 * i + 1) / (2 * n)) for i in range(n)]
vals = [g(x) for x in nodes]
coef = list(vals)
for j in range(1, n):
    for i in range(n - 1, j - 1, -1):
        coef[i] = (coef[i] - coef[i - 1]) / (nodes[i] - nodes[i - j])
poly = [D(0)] * n
poly[0] = coef[n - 1]
cur = 0
for i in range(n - 2, -1, -1):
    new = [D(0)] * n
    for k in range(cur + 1):
        new[k + 1] += poly[k]
        new[k] -= poly[k] * nodes[i]
    new[0] += coef[i]
    poly = new
    cur += 1
cs = [float(c) for c in poly]  # c1, c2, c3
for k, c in enumerate(cs):
    print(f"  c{k + 1} = {c!r}   ({c.hex()})")


def fma(a, b, c):
    return float(F(a) * F(b) + F(c))  # exact product-sum, one rounding


def rnd(x):
    return float(x)


tab = [float((D(j) * LN2 / NTAB).exp()) for j in range(NTAB)]
worst = D(0)
random.seed(1)
for trial in range(20000):
    t = random.uniform(-60, 60) if trial % 2 else random.uniform(-1.5, 1.5)
    t1024 = t * NTAB  # exact in binary (power of two scale)
    kf = float(round(t1024)) if abs(t1024 - math.floor(t1024) - 0.5) > 1e-12 else float(math.floor(t1024 / 2 + 0.5) * 2) if False else float(round(t1024))
    f = t1024 - kf  # exact
    k = int(kf)
    e, j = k >> int(math.log2(NTAB)), k & (NTAB - 1)
    p = cs[-1]
    for c in reversed(cs[:-1]):
        p = fma(p, f, c)
    p = fma(p, f, 1.0)
    val = math.ldexp(tab[j] * p, e)  # one rounding in the product
    ref = (D(t1024) * LN2 / NTAB).exp()
    err = abs(D(val) / ref - 1)
    worst = max(worst, err)
print(f"table {NTAB}, degree {DEG}: worst relative error of 2^t over 20000 points: {float(worst):.3e}")
if len(sys.argv) > 3:
    print("first entries:", [x.hex() for x in tab[:3]])
