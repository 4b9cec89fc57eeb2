#!/usr/bin/env python3
"""
Near-minimax polynomial for exp on [-ln2/2, ln2/2]: interpolation at Chebyshev nodes computed in 60-digit
decimal arithmetic, coefficients rounded to double.  Prints C literals and the measured max relative error.
"""
import sys
from decimal import Decimal as D, getcontext
from fractions import Fraction

getcontext().prec = 60
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 10
LN2 = D(2).ln()
h = LN2 / 2
import math

def cos_dec(x):
    # Taylor series for cos in Decimal
    x = D(x)
    s, term, k = D(0), D(1), 0
    while abs(term) > D(10) ** -58:
        s += term
        k += 2
        term = -term * x * x / (k * (k - 1))
    return s

PI = D("3.14159265358979323846264338327950288419716939937510582097494")
n = deg + 1
nodes = [h * cos_dec(PI * (2 * i + 1) / (2 * n)) for i in range(n)]
vals = [x.exp() for x in nodes]
# Newton divided differences
coef = list(vals)
for j in range(1, n):
    for i in range(n - 1, j - 1, -1):
        coef[i] = (coef[i] - coef[i - 1]) / (nodes[i] - nodes[i - j])
# expand to monomial basis
poly = [D(0)] * n
poly[0] = coef[n - 1]
cur_deg = 0
for i in range(n - 2, -1, -1):
    # poly = poly * (x - nodes[i]) + coef[i]
    new = [D(0)] * n
    for k in range(cur_deg + 1):
        new[k + 1] += poly[k]
        new[k] -= poly[k] * nodes[i]
    new[0] += coef[i]
    poly = new
    cur_deg += 1
cd = [float(c) for c in poly]
print("degree", deg)
for k, c in enumerate(cd):
    print(f"  c{k} = {c!r}   ({c.hex()})")
# error check in Decimal at many points using the ROUNDED coefficients evaluated exactly
worst = D(0)
M = 4001
for i in range(M):
    x = -h + (2 * h) * D(i) / D(M - 1)
    p = D(0)
    for c in reversed(cd):
        p = p * x + D(c)
    err = abs(p / x.exp() - 1)
    worst = max(worst, err)
print("max rel error with double coefficients (exact Horner):", float(worst))

# base-2 variant: 2^f on [-1/2, 1/2] (exp2 domain: f = t - rint(t), t = u * log2(e))
if len(sys.argv) > 2 and sys.argv[2] == "exp2":
    h2 = D("0.5")
    nodes = [h2 * cos_dec(PI * (2 * i + 1) / (2 * n)) for i in range(n)]
    vals = [(x * LN2).exp() for x in nodes]
    coef = list(vals)
    for jj in range(1, n):
        for i in range(n - 1, jj - 1, -1):
            coef[i] = (coef[i] - coef[i - 1]) / (nodes[i] - nodes[i - jj])
    poly = [D(0)] * n
    poly[0] = coef[n - 1]
    cur_deg = 0
    for i in range(n - 2, -1, -1):
        new = [D(0)] * n
        for k in range(cur_deg + 1):
            new[k + 1] += poly[k]
            new[k] -= poly[k] * nodes[i]
        new[0] += coef[i]
        poly = new
        cur_deg += 1
    cd = [float(c) for c in poly]
    print("exp2 degree", deg)
    for k, c in enumerate(cd):
        print(f"  c{k} = {c!r}")
    worst = D(0)
    for i in range(4001):
        x = -h2 + D(i) / D(4000)
        p = D(0)
        for c in reversed(cd):
            p = p * x + D(c)
        worst = max(worst, abs(p / (x * LN2).exp() - 1))
    print("exp2 max rel error with double coefficients (exact Horner):", float(worst))
