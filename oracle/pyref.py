"""oracle/pyref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Big-integer, definition-level restatement of the arithmetic behind the jf-plonk
prover's hot path.  Everything here is the *mathematical definition* the
arkworks crates implement (ark-ff / ark-poly / ark-ec 0.4.x, which are NOT
vendored under /root/reference -- see SURVEY.md section 8(c)); the reference
call sites the definitions are anchored on are:

  * forward / inverse / coset NTT : plonk/src/proof_system/prover.rs:54-62,545-567,672
                                    relation/src/constraint_system.rs:1162-1259
  * MSM (msm_bigint)              : primitives/src/pcs/univariate_kzg/mod.rs:109-111,151-155
  * commit (skip zeros, bigints)  : primitives/src/pcs/univariate_kzg/mod.rs:90-116,379-395
  * testing SRS  [beta^i]G        : primitives/src/pcs/univariate_kzg/srs.rs:118-153

PARITY UNPINNED: the reference holds no golden vectors / KATs for NTT, MSM,
commitments or proofs, and no Rust toolchain exists here, so
this oracle is pinned only by algebra (unique results) and by the constants of
SURVEY.md Appendix A, which are re-asserted in `self_check()`.

Pure-Python loops: small cases only (N <= 2^12 or so).
"""
from __future__ import annotations

import random
from dataclasses import dataclass


@dataclass(frozen=True)
class Curve:
    name: str
    curve_id: int
    r: int            # scalar field modulus
    q: int            # base field modulus
    fr_gen: int       # Fr::GENERATOR (multiplicative generator; coset offset prover.rs:545)
    two_adicity: int
    b: int            # y^2 = x^3 + b
    gx: int
    gy: int
    fq_limbs: int     # 64-bit limbs of Fq
    fr_limbs: int = 4

    @property
    def fq_R(self):
        return 1 << (64 * self.fq_limbs)

    @property
    def fr_R(self):
        return 1 << 256

    def root_of_unity(self, log_n: int) -> int:
        """omega_N for N = 2^log_n: (g^((r-1)/2^s))^(2^(s-log_n))  (SURVEY Appendix A)."""
        assert 0 <= log_n <= self.two_adicity
        root = pow(self.fr_gen, (self.r - 1) >> self.two_adicity, self.r)
        return pow(root, 1 << (self.two_adicity - log_n), self.r)


BLS12_381 = Curve(
    name="bls12-381", curve_id=0,
    r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    fr_gen=7, two_adicity=32, b=4,
    gx=0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    gy=0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
    fq_limbs=6)

BN254 = Curve(
    name="bn254", curve_id=1,
    r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
    q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
    fr_gen=5, two_adicity=28, b=3, gx=1, gy=2, fq_limbs=4)

CURVES = {0: BLS12_381, 1: BN254}


# --------------------------------------------------------------------------
# limb encodings (Appendix B: N little-endian u64 limbs holding a*R mod p)
# --------------------------------------------------------------------------
def to_limbs(x: int, n_limbs: int) -> list[int]:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n_limbs)]


def from_limbs(limbs) -> int:
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def fr_to_mont(c: Curve, x: int) -> int:
    return (x * c.fr_R) % c.r


def fr_from_mont(c: Curve, x: int) -> int:
    return (x * pow(c.fr_R, -1, c.r)) % c.r


def fq_to_mont(c: Curve, x: int) -> int:
    return (x * c.fq_R) % c.q


def fq_from_mont(c: Curve, x: int) -> int:
    return (x * pow(c.fq_R, -1, c.q)) % c.q


# --------------------------------------------------------------------------
# NTT by definition (Appendix B)
# --------------------------------------------------------------------------
def ntt_def(c: Curve, coeffs: list[int], log_n: int, offset: int = 1) -> list[int]:
    """out[i] = sum_j c[j] * (offset * w^i)^j, natural order; input zero-padded to N."""
    n = 1 << log_n
    assert len(coeffs) <= n
    w = c.root_of_unity(log_n)
    out = []
    for i in range(n):
        x = offset * pow(w, i, c.r) % c.r
        acc = 0
        for cj in reversed(coeffs):       # Horner
            acc = (acc * x + cj) % c.r
        out.append(acc)
    return out


def intt_def(c: Curve, evals: list[int], log_n: int, offset: int = 1) -> list[int]:
    """c[j] = offset^-j * N^-1 * sum_i e[i] * w^(-i*j), length N."""
    n = 1 << log_n
    assert len(evals) <= n
    evals = list(evals) + [0] * (n - len(evals))
    w_inv = pow(c.root_of_unity(log_n), -1, c.r)
    n_inv = pow(n, -1, c.r)
    off_inv = pow(offset, -1, c.r)
    out = []
    for j in range(n):
        x = pow(w_inv, j, c.r)
        acc = 0
        for e in reversed(evals):
            acc = (acc * x + e) % c.r
        out.append(acc * n_inv % c.r * pow(off_inv, j, c.r) % c.r)
    return out


def ntt_fast(c: Curve, a: list[int], log_n: int, offset: int = 1, inverse: bool = False) -> list[int]:
    """O(N log N) recursive radix-2, same definition as ntt_def / intt_def (mid sizes)."""
    n = 1 << log_n
    r = c.r
    a = list(a) + [0] * (n - len(a))
    w = c.root_of_unity(log_n)
    if inverse:
        w = pow(w, -1, r)
    else:
        p = 1
        for j in range(n):
            a[j] = a[j] * p % r
            p = p * offset % r

    def rec(v, wn):
        m = len(v)
        if m == 1:
            return v
        e = rec(v[0::2], wn * wn % r)
        o = rec(v[1::2], wn * wn % r)
        out = [0] * m
        t = 1
        h = m // 2
        for k in range(h):
            x = o[k] * t % r
            out[k] = (e[k] + x) % r
            out[k + h] = (e[k] - x) % r
            t = t * wn % r
        return out

    out = rec(a, w)
    if inverse:
        n_inv = pow(n, -1, r)
        off_inv = pow(offset, -1, r)
        p = n_inv
        for j in range(n):
            out[j] = out[j] * p % r
            p = p * off_inv % r
    return out


# --------------------------------------------------------------------------
# G1 arithmetic (affine, None = infinity), a = 0 short Weierstrass
# --------------------------------------------------------------------------
def g1_on_curve(c: Curve, P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - c.b) % c.q == 0


def g1_neg(c: Curve, P):
    return None if P is None else (P[0], (-P[1]) % c.q)


def g1_add(c: Curve, P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    q = c.q
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
    x3 = (lam * lam - x1 - x2) % q
    y3 = (lam * (x1 - x3) - y1) % q
    return (x3, y3)


def g1_mul(c: Curve, k: int, P):
    """double-and-add; k is taken as a plain non-negative integer (msm_bigint semantics)."""
    R = None
    A = P
    while k:
        if k & 1:
            R = g1_add(c, R, A)
        A = g1_add(c, A, A)
        k >>= 1
    return R


def g1_gen(c: Curve):
    return (c.gx, c.gy)


def msm_def(c: Curve, bases, scalars):
    """sum_i k_i * P_i over the first min(len) pairs (Appendix B; mod.rs:109-111)."""
    acc = None
    for P, k in zip(bases, scalars):
        acc = g1_add(c, acc, g1_mul(c, k, P))
    return acc


def jacobian_to_affine(c: Curve, X: int, Y: int, Z: int):
    if Z % c.q == 0:
        return None
    zi = pow(Z, -1, c.q)
    return (X * zi * zi % c.q, Y * zi * zi * zi % c.q)


def srs_powers(c: Curve, beta: int, n: int):
    """[beta^i]G for i < n  (gen_srs_for_testing, srs.rs:118-153, with g = generator)."""
    G = g1_gen(c)
    out = []
    p = 1
    for _ in range(n):
        out.append(g1_mul(c, p, G))
        p = p * beta % c.r
    return out


def poly_eval(c: Curve, coeffs, x: int) -> int:
    acc = 0
    for cj in reversed(coeffs):
        acc = (acc * x + cj) % c.r
    return acc


# --------------------------------------------------------------------------
# arkworks window rule, only to label the CPU baseline
# --------------------------------------------------------------------------
def ark_window_bits(n: int) -> int:
    if n < 32:
        return 3
    log2_ceil = (n - 1).bit_length()
    return log2_ceil * 69 // 100 + 2


def self_check():
    for c in (BLS12_381, BN254):
        assert pow(c.fr_gen, (c.r - 1) // 2, c.r) == c.r - 1
        assert (c.r - 1) % (1 << c.two_adicity) == 0 and ((c.r - 1) >> c.two_adicity) & 1
        w = c.root_of_unity(c.two_adicity)
        assert pow(w, 1 << (c.two_adicity - 1), c.r) == c.r - 1
        assert g1_on_curve(c, g1_gen(c))
        assert g1_mul(c, c.r, g1_gen(c)) is None          # generator has order r
    assert BLS12_381.root_of_unity(32) == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    assert BN254.root_of_unity(28) == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert BLS12_381.fr_R % BLS12_381.r == 0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe
    assert BN254.fr_R % BN254.r == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
    assert [ark_window_bits(1 << k) for k in (10, 15, 20, 22)] == [8, 12, 15, 17]
    assert ark_window_bits((1 << 20) + 2) == 16
    # definition vs fast NTT
    rng = random.Random(1)
    for c in (BLS12_381, BN254):
        a = [rng.randrange(c.r) for _ in range(13)]
        assert ntt_def(c, a, 4, c.fr_gen) == ntt_fast(c, a, 4, c.fr_gen)
        e = ntt_def(c, a, 4, c.fr_gen)
        assert intt_def(c, e, 4, c.fr_gen) == a + [0, 0, 0] == ntt_fast(c, e, 4, c.fr_gen, inverse=True)
    return True


if __name__ == "__main__":
    print("pyref self-check:", self_check())
