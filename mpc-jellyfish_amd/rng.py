"""The reference's deterministic randomness, host side:

    jf_utils::test_rng()                    utilities/src/lib.rs:62-70     StdRng (= ChaCha12, rand 0.8) from a fixed seed
    compute_coset_representatives           relation/src/constants.rs:30-80  ChaChaRng (= ChaCha20) from the all-zero seed
    F::rand / DensePolynomial::rand         [upstream ark-ff 0.4 / ark-poly 0.4] as used by mask_polynomial (prover.rs:463-486)
                                            and split_quotient_polynomial (prover.rs:946-957)

rand_chacha: the 32-byte seed is the key, 64-bit block counter in state words 12-13, stream id 0 in words 14-15,
output consumed as little-endian u32 words in block order.  `next_u64` = two consecutive words, low word first
(rand_core::block::BlockRng; the index stays even when only u64s are drawn).
ark-ff `Fp::rand`: draw N u64 limbs (least significant first), clear the unused top bits of the last limb, accept if
the integer is below the modulus -- and the limbs ARE the element's Montgomery representation.
"""
from __future__ import annotations

import struct

from .params import CurveParams, curve as _curve

_M32 = 0xFFFFFFFF
TEST_RNG_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)      # utilities/src/lib.rs:65-68


def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & _M32


def chacha_block(key_words, counter: int, rounds: int, stream: int = 0):
    """One 64-byte ChaCha block as 16 u32 words (djb variant: 64-bit counter, 64-bit stream id)."""
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & _M32, (counter >> 32) & _M32,
                                                                                 stream & _M32, (stream >> 32) & _M32]
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + init[i]) & _M32 for i in range(16)]


class ChaChaRng:
    """rand_chacha::ChaCha{8,12,20}Rng::from_seed(seed)."""

    def __init__(self, seed: bytes, rounds: int):
        assert len(seed) == 32 and rounds in (8, 12, 20)
        self.key = struct.unpack("<8I", seed)
        self.rounds = rounds
        self.counter = 0
        self.buf: list[int] = []
        self.index = 0

    def _refill(self):
        """rand_chacha produces four blocks per refill; the block function runs in the library (host-only mzk_chacha_blocks:
        pure Python costs 0.5 ms of every proof's blinding draws); `chacha_block` above is the same function, kept as its
        definition and checked against it by the tests."""
        import ctypes as C
        from . import lib as _lib
        key = (C.c_uint32 * 8)(*self.key)
        out = (C.c_uint32 * 64)()
        _lib.check(_lib.load().mzk_chacha_blocks(C.addressof(key), self.counter, self.rounds, 4, C.addressof(out)), "mzk_chacha_blocks")
        self.buf = list(out)
        self.counter += 4
        self.index = 0

    def next_u32(self) -> int:
        if self.index >= len(self.buf):
            self._refill()
        v = self.buf[self.index]
        self.index += 1
        return v

    def next_u64(self) -> int:
        n = len(self.buf)
        if self.index < n - 1:
            lo, hi = self.buf[self.index], self.buf[self.index + 1]
            self.index += 2
            return (hi << 32) | lo
        if self.index >= n:
            self._refill()
            self.index = 2
            return (self.buf[1] << 32) | self.buf[0]
        lo = self.buf[n - 1]                                 # odd index at the end of the buffer (after a lone next_u32)
        self._refill()
        self.index = 1
        return (self.buf[0] << 32) | lo


def test_rng() -> ChaChaRng:
    """jf_utils::test_rng (utilities/src/lib.rs:62-70)."""
    return ChaChaRng(TEST_RNG_SEED, 12)


def _mont_r_inv(c: CurveParams) -> int:
    return pow(1 << 256, -1, c.r)


def fr_rand(curve, rng: ChaChaRng) -> int:
    """`Fr::rand(rng)`: returns the canonical value of the element whose Montgomery limbs were drawn."""
    c = _curve(curve)
    shave = 256 - c.r.bit_length()
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= (1 << (64 - shave)) - 1
        v = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
        if v < c.r:
            return v * c.fr_Rinv % c.r


def dense_poly_rand(curve, degree: int, rng: ChaChaRng) -> list[int]:
    """`DensePolynomial::rand(degree, rng)`: degree + 1 coefficients, low order first."""
    return [fr_rand(curve, rng) for _ in range(degree + 1)]


def compute_coset_representatives(curve, num_wire_types: int, coset_size: int | None = None) -> list[int]:
    """relation/src/constants.rs:30-80: k_0 = 1, then ChaCha20(seed 0) draws, rejecting a k whose coset k*H repeats."""
    c = _curve(curve)
    rng = ChaChaRng(bytes(32), 20)
    n = coset_size if coset_size is not None else 1 << c.two_adicity
    ks, pows = [1], [1]
    for _ in range(1, num_wire_types):
        while True:
            k = fr_rand(c, rng)
            p = pow(k, n, c.r)
            if p not in pows:                                # (a^-1 b)^N == 1  <=>  a^N == b^N
                break
        ks.append(k)
        pows.append(p)
    return ks


# ---- `universal_setup_for_testing` (plonk/src/proof_system/snark.rs:485-526; primitives/src/pcs/univariate_kzg/srs.rs:118-153) ----------
# The reference's tests and its bench build their SRS with beta = Fr::rand, g = G1::rand, h = G2::rand drawn from the SAME rng that
# `prove` then takes its blinders from.  A host that wants the reference's proof bytes for such a setup mirrors those draws:
# [upstream ark-ec 0.4] `impl Distribution<Projective<P>> for Standard`: loop { x = BaseField::rand; greatest = rng.gen::<bool>()
# (rand 0.8: the top bit of one u32); get_point_from_x_unchecked(x, greatest) -- the two roots of x^3 + b ordered as integers --
# or draw again }, then the point times P::COFACTOR.
G1_COFACTOR = {0: 0x396c8c005555e1568c00aaab0000aaab, 1: 1}


def fq_rand(curve, rng: ChaChaRng) -> int:
    c = _curve(curve)
    nl = c.fq_limbs
    shave = 64 * nl - c.q.bit_length()
    while True:
        limbs = [rng.next_u64() for _ in range(nl)]
        limbs[-1] &= (1 << (64 - shave)) - 1
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < c.q:
            return v * pow(1 << (64 * nl), -1, c.q) % c.q


def g1_rand(curve, rng: ChaChaRng):
    """`E::G1::rand(rng)`: affine (x, y) as canonical integers.  The cofactor multiple is one single-pair MSM on the device."""
    import numpy as np
    from . import kzg
    from .params import fq_from_mont, fq_to_mont, fr_bigints
    c = _curve(curve)
    b = 4 if c.curve_id == 0 else 3
    while True:
        x = fq_rand(c, rng)
        greatest = rng.next_u32() >> 31 == 1
        rhs = (x * x % c.q * x + b) % c.q
        y = pow(rhs, (c.q + 1) // 4, c.q)                     # q = 3 mod 4 on both curves
        if y * y % c.q != rhs:
            continue
        y = max(y, c.q - y) if greatest else min(y, c.q - y)
        if G1_COFACTOR[c.curve_id] == 1:
            return (x, y)
        pt = np.concatenate([fq_to_mont(c, [x])[0], fq_to_mont(c, [y])[0]]).reshape(1, -1)
        pp = kzg.UnivariateProverParam.from_affine(c, pt)
        aff = kzg.jacobian_to_affine(c, kzg.msm_bigint(pp, fr_bigints([G1_COFACTOR[c.curve_id]]))[None])[0]
        pp.release()
        return tuple(fq_from_mont(c, aff))


def g2_rand_skip(curve, rng: ChaChaRng) -> None:
    """`E::G2::rand(rng)` as far as a prover is concerned: the draws it takes (h sits in the verifying key's open key only).
    Fq2 = Fq[u] / (u^2 + 1); x^3 + b' (4 (1 + u) on BLS12-381, 3 / (9 + u) on BN254) is a square iff its norm is one in Fq."""
    c = _curve(curve)
    q = c.q
    mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)
    b2 = (4, 4) if c.curve_id == 0 else (27 * pow(82, -1, q) % q, (-3) * pow(82, -1, q) % q)
    while True:
        x = (fq_rand(c, rng), fq_rand(c, rng))
        rng.next_u32()
        x3 = mul(mul(x, x), x)
        rhs = ((x3[0] + b2[0]) % q, (x3[1] + b2[1]) % q)
        norm = (rhs[0] * rhs[0] + rhs[1] * rhs[1]) % q
        if norm == 0 or pow(norm, (q - 1) // 2, q) == 1:
            return


def universal_setup_for_testing(curve, rng: ChaChaRng):
    """(beta, g): powers_of_g[i] = beta^i g (kzg.UnivariateProverParam.gen_srs_for_testing(c, beta, degree, g=g)); the rng is left
    where `prove` finds it after the reference's setup."""
    beta = fr_rand(curve, rng)
    g = g1_rand(curve, rng)
    g2_rand_skip(curve, rng)
    return beta, g
