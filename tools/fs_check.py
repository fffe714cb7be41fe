#!/usr/bin/env python3
"""tools/fs_check.py -- verifies tools/fs_check.cpp's output with big integers: the plan-time twiddle records (w, wq), the
constant-operand Barrett product fs_mulc (congruence, range (-2p, 3p), limb class) and fs_canonical, on random, signed-extreme and
edge inputs.  Host-only: the same header compiles for gfx950.

    g++ -O2 -std=c++17 -I mpc-jellyfish_amd/csrc tools/fs_check.cpp -o /tmp/fs_check && /tmp/fs_check 20000 | python tools/fs_check.py
"""
import sys

P = {"bls": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
     "bn": 21888242871839275222246405745257275088548364400416034343698204186575808495617}
val = lambda limbs: sum(int(l) << (29 * i) for i, l in enumerate(limbs))
n = 0
worst_lo, worst_hi = 0.0, 0.0
for line in sys.stdin:
    f = line.split()
    p = P[f[0]]
    v = [int(x) for x in f[1:]]
    w, q, x, r, cx, cr = (v[9 * i:9 * i + 9] for i in range(6))
    W, Q, X, R = val(w), val(q), val(x), val(r)
    assert 0 <= W < p and all(0 <= l < 1 << 29 for l in w + q), "twiddle limbs"
    assert Q == (W << 261) // p, "wq != floor(w 2^261 / p)"
    assert abs(X) < 1 << 261
    assert (R - X * W) % p == 0, "fs_mulc not congruent"
    assert -2 * p < R < 3 * p, ("fs_mulc range", R / p)
    assert all(0 <= l < 1 << 29 for l in r[:8]) and -(1 << 28) <= r[8] < 1 << 28, "class C"
    assert val(cx) == X % p and val(cr) == (X * W) % p and all(0 <= l < 1 << 29 for l in cx + cr), "fs_canonical"
    worst_lo, worst_hi = min(worst_lo, R / p), max(worst_hi, R / p)
    n += 1
print("fs_check: %d cases ok; fs_mulc results within (%.3f p, %.3f p)" % (n, worst_lo, worst_hi))
