"""oracle/pyref_pairing.py -- TEST INFRASTRUCTURE ONLY.

Ate pairings on BLS12-381 and BN254 by the definition -- extension fields as polynomial quotient rings, G2 points moved to
E(Fq12) through the twist isomorphism, an affine Miller loop, the final exponentiation as one big power -- so that the
restated verifier (pyref_verifier.batch_verify_opening_proofs) can evaluate the reference's own check
    multi_pairing([A, -B], [beta_h, h]) == 1          plonk/src/proof_system/verifier.rs:226-250
    e(C - [v]G, h) == e(proof, beta_h - [z]h)         primitives/src/pcs/univariate_kzg/mod.rs:194-221
instead of its trapdoor form (beta A == B).  Slow (seconds per pairing) and used by a handful of tests only.

Everything is self-checking (`self_check`): the G2 generators lie on their twist curves and have order r, the twist lands on
y^2 = x^3 + b over Fq12, and the pairing is bilinear and non-degenerate.  The towers follow the standard construction
(Fq2 = Fq[u]/(u^2+1); Fq12 = Fq[w]/(w^12 - 18 w^6 + 82) for BN254 and Fq[w]/(w^12 - 2 w^6 + 2) for BLS12-381, i.e. w^6 = 9 + u
resp. 1 + u).  The sign of the BLS12-381 loop parameter is ignored: that conjugates every pairing value alike and leaves
product checks (== 1) and bilinearity untouched.
"""
from __future__ import annotations

import pyref as P


class Ext:
    """Element of Fq[X] / (X^deg + sum mod_coeffs[i] X^i): coefficient tuple, low degree first."""
    __slots__ = ("c", "f")

    def __init__(self, field, coeffs):
        self.f = field
        self.c = tuple(x % field.q for x in coeffs)

    def __eq__(self, o):
        return self.c == o.c

    def __hash__(self):
        return hash(self.c)

    def __add__(self, o):
        return Ext(self.f, [a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return Ext(self.f, [a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return Ext(self.f, [-a for a in self.c])

    def scale(self, k: int):
        return Ext(self.f, [a * k for a in self.c])

    def __mul__(self, o):
        f = self.f
        d, q = f.deg, f.q
        t = [0] * (2 * d - 1)
        for i, a in enumerate(self.c):
            if a:
                for j, b in enumerate(o.c):
                    t[i + j] += a * b
        for k in range(2 * d - 2, d - 1, -1):                 # X^k = -sum mod[i] X^(k-d+i)
            top = t[k] % q
            if top:
                for i, m in f.sparse_mod:
                    t[k - d + i] -= top * m
        return Ext(f, t[:d])

    def is_zero(self):
        return not any(self.c)

    def inv(self):
        """extended Euclid in Fq[X] against the modulus polynomial"""
        f = self.f
        q, d = f.q, f.deg
        lm, hm = [1] + [0] * d, [0] * (d + 1)
        low, high = list(self.c) + [0], list(f.mod_full)
        deg = lambda p: max((i for i, x in enumerate(p) if x), default=0)
        while deg(low):
            dl, dh = deg(low), deg(high)
            # r = high / low (polynomial division, quotient only)
            r = [0] * (d + 1)
            temp = list(high)
            inv_lead = pow(low[dl], -1, q)
            for i in range(dh - dl, -1, -1):
                r[i] = temp[dl + i] * inv_lead % q
                if r[i]:
                    for j in range(dl + 1):
                        temp[i + j] = (temp[i + j] - r[i] * low[j]) % q
            nm, new = list(hm), list(high)
            for i in range(d + 1):
                for j in range(d + 1 - i):
                    nm[i + j] -= lm[i] * r[j]
                    new[i + j] -= low[i] * r[j]
            nm = [x % q for x in nm]
            new = [x % q for x in new]
            lm, low, hm, high = nm, new, lm, low
        k = pow(low[0], -1, q)
        return Ext(f, [x * k for x in lm[:d]])

    def __truediv__(self, o):
        return self * o.inv()

    def __pow__(self, e: int):
        out, base = self.f.one(), self
        while e:
            if e & 1:
                out = out * base
            base = base * base
            e >>= 1
        return out


class ExtField:
    def __init__(self, q: int, mod_coeffs):
        self.q, self.deg = q, len(mod_coeffs)
        self.mod_full = [m % q for m in mod_coeffs] + [1]
        self.sparse_mod = [(i, m) for i, m in enumerate(mod_coeffs) if m]

    def __call__(self, coeffs):
        return Ext(self, list(coeffs) + [0] * (self.deg - len(coeffs)))

    def one(self):
        return self([1])

    def zero(self):
        return self([])


# --- affine short-Weierstrass arithmetic over any of the fields above (None = infinity) ---------------------------------------
def ec_double(pt):
    if pt is None:
        return None
    x, y = pt
    if y.is_zero():
        return None
    lam = (x * x).scale(3) / y.scale(2)
    nx = lam * lam - x - x
    return nx, lam * (x - nx) - y


def ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        return ec_double(p1) if y1 == y2 else None
    lam = (y2 - y1) / (x2 - x1)
    nx = lam * lam - x1 - x2
    return nx, lam * (x1 - nx) - y1


def ec_mul(pt, k: int):
    out = None
    while k:
        if k & 1:
            out = ec_add(out, pt)
        pt = ec_double(pt)
        k >>= 1
    return out


def ec_neg(pt):
    return None if pt is None else (pt[0], -pt[1])


class PairingCurve:
    def __init__(self, c, xi, fq12_mod, ate_loop_count, twist_divides: bool, bn_frobenius_steps: bool, g2):
        self.c = c
        self.Fq2 = ExtField(c.q, [1, 0])                                   # u^2 + 1
        self.Fq12 = ExtField(c.q, fq12_mod)
        self.xi = xi                                                       # w^6 = xi[0] + xi[1] u
        self.ate, self.twist_divides, self.bn_steps = ate_loop_count, twist_divides, bn_frobenius_steps
        xi2 = self.Fq2(xi)
        b = self.Fq2([c.b])
        self.b2 = b * xi2 if twist_divides else b / xi2                   # the sextic twist E': y^2 = x^3 + b2
        self.w = self.Fq12([0, 1])
        self.g2 = (self.Fq2(g2[0]), self.Fq2(g2[1]))

    def on_twist(self, q) -> bool:
        return q is None or q[1] * q[1] == q[0] * q[0] * q[0] + self.b2

    def twist(self, q):
        """E'(Fq2) -> E(Fq12): write a + b u with u = w^6 - xi0 (so that the image lies in Fq[w^6]), then scale by powers of w"""
        if q is None:
            return None
        x, y = q
        F = self.Fq12
        xi0, xi1 = self.xi
        assert xi1 == 1
        nx = F([x.c[0] - x.c[1] * xi0] + [0] * 5 + [x.c[1]])
        ny = F([y.c[0] - y.c[1] * xi0] + [0] * 5 + [y.c[1]])
        w2, w3 = self.w * self.w, self.w * self.w * self.w
        return (nx / w2, ny / w3) if self.twist_divides else (nx * w2, ny * w3)

    def cast_g1(self, p):
        return None if p is None else (self.Fq12([p[0]]), self.Fq12([p[1]]))

    @staticmethod
    def _line(p1, p2, t):
        """the line through p1 and p2 (tangent if equal) evaluated at t"""
        x1, y1 = p1
        x2, y2 = p2
        xt, yt = t
        if x1 != x2:
            m = (y2 - y1) / (x2 - x1)
            return m * (xt - x1) - (yt - y1)
        if y1 == y2:
            m = (x1 * x1).scale(3) / y1.scale(2)
            return m * (xt - x1) - (yt - y1)
        return xt - x1

    def miller_loop(self, q2, p1):
        """f_{ate,Q}(P) before the final exponentiation; q2 on the twist over Fq2, p1 an affine G1 point"""
        if q2 is None or p1 is None:
            return self.Fq12.one()
        Q, Pt = self.twist(q2), self.cast_g1(p1)
        R, f = Q, self.Fq12.one()
        for i in range(self.ate.bit_length() - 2, -1, -1):
            f = f * f * self._line(R, R, Pt)
            R = ec_double(R)
            if (self.ate >> i) & 1:
                f = f * self._line(R, Q, Pt)
                R = ec_add(R, Q)
        if self.bn_steps:                                                  # BN curves: two more lines through the Frobenius images of Q
            q = self.c.q
            Q1 = (Q[0] ** q, Q[1] ** q)
            nQ2 = (Q1[0] ** q, -(Q1[1] ** q))
            f = f * self._line(R, Q1, Pt)
            R = ec_add(R, Q1)
            f = f * self._line(R, nQ2, Pt)
        return f

    def final_exponentiation(self, f):
        return f ** ((self.c.q ** 12 - 1) // self.c.r)

    def pairing(self, q2, p1):
        return self.final_exponentiation(self.miller_loop(q2, p1))

    def multi_pairing_is_one(self, pairs) -> bool:
        """prod e(P_i, Q_i) == 1 for pairs (P_i in G1, Q_i in G2): one final exponentiation over the product of the Miller loops"""
        f = self.Fq12.one()
        for p1, q2 in pairs:
            f = f * self.miller_loop(q2, p1)
        return self.final_exponentiation(f) == self.Fq12.one()

    def g2_mul(self, q2, k: int):
        return ec_mul(q2, k % self.c.r)


BN254_PAIRING = PairingCurve(
    P.BN254, xi=(9, 1), fq12_mod=[82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0], ate_loop_count=29793968203157093288, twist_divides=False,
    bn_frobenius_steps=True,
    g2=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531)))

BLS12_381_PAIRING = PairingCurve(
    P.BLS12_381, xi=(1, 1), fq12_mod=[2, 0, 0, 0, 0, 0, -2, 0, 0, 0, 0, 0], ate_loop_count=15132376222941642752, twist_divides=True,
    bn_frobenius_steps=False,
    g2=((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
         0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
        (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
         0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)))

PAIRINGS = {0: BLS12_381_PAIRING, 1: BN254_PAIRING}


def self_check(curve_id: int, bilinearity: bool = True):
    pc = PAIRINGS[curve_id]
    c = pc.c
    assert pc.on_twist(pc.g2), "G2 generator is not on the twist"
    assert ec_mul(pc.g2, c.r) is None and ec_mul(pc.g2, 1) is not None, "G2 generator does not have order r"
    tq = pc.twist(pc.g2)
    assert tq[1] * tq[1] == tq[0] * tq[0] * tq[0] + pc.Fq12([c.b]), "twist does not land on E(Fq12)"
    if not bilinearity:
        return
    G = P.g1_gen(c)
    e = pc.pairing(pc.g2, G)
    assert e != pc.Fq12.one() and e ** c.r == pc.Fq12.one(), "pairing degenerate or not of order r"
    a, b = 0x1234567, 0x89abcde
    assert pc.pairing(pc.g2_mul(pc.g2, b), P.g1_mul(c, a, G)) == e ** (a * b), "pairing is not bilinear"
