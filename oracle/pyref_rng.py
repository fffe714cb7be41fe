"""oracle/pyref_rng.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

The reference's deterministic randomness, restated from the crates' published definitions with NO import from the
product package (mpc-jellyfish_amd/rng.py is a separate text by design):

  jf_utils::test_rng                /root/reference/utilities/src/lib.rs:62-70    rand 0.8 `StdRng` = ChaCha12, fixed seed
  compute_coset_representatives     /root/reference/relation/src/constants.rs:30-80   `ChaChaRng` = ChaCha20, zero seed
  F::rand                           ark-ff 0.4 `impl Distribution<Fp<P, N>> for Standard` (rejection sampling on masked limbs)
  DensePolynomial::rand             ark-poly 0.4: degree + 1 draws of F::rand, low order first
  draw order of one proof           /root/reference/plonk/src/proof_system/prover.rs:79-83, 113-114, 133-138, 169-180, 947-955
  draw order of a batch             /root/reference/plonk/src/proof_system/snark.rs:277-399 (round by round, instance by instance)
  universal_setup_for_testing       /root/reference/plonk/src/proof_system/snark.rs:485-526: beta = Fr::rand, g = G1::rand, h = G2::rand
  G1::rand / G2::rand               ark-ec 0.4 `impl Distribution<Projective<P>> for Standard` [upstream, restated from the published
                                    source; pinned only when integration/rust/gen_fixtures has run]: loop { x = BaseField::rand;
                                    greatest = rng.gen::<bool>(); if x^3 + b is a square: the root chosen by `greatest`, times COFACTOR }

ChaCha (Bernstein 2008; RFC 7539 section 2.1-2.3 for the quarter round and the state layout): constants "expand 32-byte k",
key = the 32-byte seed, words 12-13 = 64-bit block counter, words 14-15 = stream id 0 (rand_chacha's layout, the original
"djb" one).  rand_chacha fills a buffer of FOUR consecutive blocks at a time and `rand_core::block::BlockRng` hands it out
as little-endian u32 words; `next_u64` takes two consecutive words (low word first) and, when only one word is left, pairs
it with the first word of the next buffer.
"""
from __future__ import annotations

import struct

M32 = 0xFFFFFFFF
SIGMA = struct.unpack("<4I", b"expand 32-byte k")
TEST_RNG_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0]) + bytes(16)          # utilities/src/lib.rs:65-68
BUF_WORDS = 64                                                                                        # four 16-word blocks


def _quarter(s, a, b, c, d):
    """RFC 7539 2.1"""
    s[a] = (s[a] + s[b]) & M32; s[d] ^= s[a]; s[d] = ((s[d] << 16) | (s[d] >> 16)) & M32
    s[c] = (s[c] + s[d]) & M32; s[b] ^= s[c]; s[b] = ((s[b] << 12) | (s[b] >> 20)) & M32
    s[a] = (s[a] + s[b]) & M32; s[d] ^= s[a]; s[d] = ((s[d] << 8) | (s[d] >> 24)) & M32
    s[c] = (s[c] + s[d]) & M32; s[b] ^= s[c]; s[b] = ((s[b] << 7) | (s[b] >> 25)) & M32


def chacha_words(seed: bytes, block: int, rounds: int) -> list:
    """The 16 output words of block number `block` under key `seed`."""
    state = list(SIGMA) + list(struct.unpack("<8I", seed)) + [block & M32, block >> 32, 0, 0]
    work = list(state)
    for _ in range(rounds // 2):
        for col in range(4):                                   # column round
            _quarter(work, col, 4 + col, 8 + col, 12 + col)
        for dg in range(4):                                    # diagonal round
            _quarter(work, dg, 4 + (dg + 1) % 4, 8 + (dg + 2) % 4, 12 + (dg + 3) % 4)
    return [(w + s) & M32 for w, s in zip(work, state)]


class BlockRng:
    """rand_core::block::BlockRng<ChaChaXCore>"""

    def __init__(self, seed: bytes, rounds: int):
        if len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        self.seed, self.rounds = seed, rounds
        self.next_block = 0
        self.words = [0] * BUF_WORDS
        self.index = BUF_WORDS                                 # empty buffer: the first draw generates

    def _generate(self, index_after: int):
        self.words = []
        for _ in range(BUF_WORDS // 16):
            self.words += chacha_words(self.seed, self.next_block, self.rounds)
            self.next_block += 1
        self.index = index_after

    def next_u32(self) -> int:
        if self.index >= BUF_WORDS:
            self._generate(0)
        v = self.words[self.index]
        self.index += 1
        return v

    def next_u64(self) -> int:
        i = self.index
        if i < BUF_WORDS - 1:
            self.index = i + 2
            return self.words[i] | (self.words[i + 1] << 32)
        if i >= BUF_WORDS:
            self._generate(2)
            return self.words[0] | (self.words[1] << 32)
        low = self.words[BUF_WORDS - 1]
        self._generate(1)
        return low | (self.words[0] << 32)


def test_rng() -> BlockRng:
    return BlockRng(TEST_RNG_SEED, 12)


def fr_rand(c, rng: BlockRng) -> int:
    """`Fr::rand`: four u64 limbs, least significant first; the top `256 - bits(r)` bits of the last limb are cleared; the
    candidate is accepted when below r -- and it is then the element's MONTGOMERY representation (the sampler fills the
    inner BigInt directly), so the canonical value is candidate / 2^256 mod r.  `c` is a pyref.Curve."""
    shave = 256 - c.r.bit_length()
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= (1 << 64) - 1 >> shave
        cand = sum(l << (64 * i) for i, l in enumerate(limbs))
        if cand < c.r:
            return cand * pow(1 << 256, -1, c.r) % c.r


def dense_poly_rand(c, degree: int, rng: BlockRng) -> list:
    return [fr_rand(c, rng) for _ in range(degree + 1)]


def compute_coset_representatives(c, num_wire_types: int, coset_size=None) -> list:
    """constants.rs:30-80: k_0 = 1; each further k is drawn from ChaCha20(zero seed) until k^N differs from every earlier
    k_i^N  ((a^-1 b)^N = 1  <=>  a^N = b^N), N = the coset size (default 2^two_adicity)."""
    rng = BlockRng(bytes(32), 20)
    big_n = coset_size if coset_size is not None else 1 << c.two_adicity
    ks, pows = [1], [1]
    while len(ks) < num_wire_types:
        k = fr_rand(c, rng)
        kn = pow(k, big_n, c.r)
        if kn in pows:
            continue
        ks.append(k)
        pows.append(kn)
    return ks


def draw_blinders(c, rng: BlockRng, num_wire_types: int, ultra: bool) -> dict:
    """Every random draw of ONE proof in the order `batch_prove_internal` makes them for a single instance:
    round 1 `mask_polynomial(rng, wire, 1)` per wire (prover.rs:79-83, hiding degree 1 -> DensePolynomial::rand(1) = 2 draws),
    round 1.5 h_1, h_2 with hiding degree 2 (:113-114, 3 draws each), round 2 z (:133-138, 3 draws), round 2.5 the lookup
    product (:169-180, 3 draws), round 3 one F::rand per split-quotient boundary (:947-955, num_wire_types - 1 draws)."""
    wires = [dense_poly_rand(c, 1, rng) for _ in range(num_wire_types)]
    h = [dense_poly_rand(c, 2, rng) for _ in range(2)] if ultra else None
    z = dense_poly_rand(c, 2, rng)
    pl = dense_poly_rand(c, 2, rng) if ultra else None
    quot = [fr_rand(c, rng) for _ in range(num_wire_types - 1)]
    return {"wires": wires, "z": z, "quot": quot, "h": h, "prod_lookup": pl}


def draw_batch_blinders(c, rng: BlockRng, num_wire_types: int, ultra_flags) -> tuple:
    """The draws of `batch_prove_internal` over several instances (snark.rs:277-399): each round loops over the instances
    before the next round starts, and the split-quotient masks are drawn once for the aggregated quotient."""
    per = [{"wires": None, "z": None, "h": None, "prod_lookup": None} for _ in ultra_flags]
    for b in per:
        b["wires"] = [dense_poly_rand(c, 1, rng) for _ in range(num_wire_types)]
    for b, u in zip(per, ultra_flags):
        if u:
            b["h"] = [dense_poly_rand(c, 2, rng) for _ in range(2)]
    for b in per:
        b["z"] = dense_poly_rand(c, 2, rng)
    for b, u in zip(per, ultra_flags):
        if u:
            b["prod_lookup"] = dense_poly_rand(c, 2, rng)
    quot = [fr_rand(c, rng) for _ in range(num_wire_types - 1)]
    return per, quot


# ---- universal_setup_for_testing (snark.rs:485-526; primitives/src/pcs/univariate_kzg/srs.rs:118-153) ----------------------------
G1_COFACTOR = {0: 0x396c8c005555e1568c00aaab0000aaab, 1: 1}                 # ark-bls12-381 / ark-bn254 `G1Config::COFACTOR`


def fq_rand(c, rng: BlockRng) -> int:
    """`Fq::rand`, as fr_rand: fq_limbs u64 draws, top bits shaved, accepted below q; the limbs are the Montgomery image."""
    n = c.fq_limbs
    shave = 64 * n - c.q.bit_length()
    while True:
        limbs = [rng.next_u64() for _ in range(n)]
        limbs[n - 1] &= ((1 << 64) - 1) >> shave
        cand = sum(l << (64 * i) for i, l in enumerate(limbs))
        if cand < c.q:
            return cand * pow(1 << (64 * n), -1, c.q) % c.q


def gen_bool(rng: BlockRng) -> bool:
    """rand 0.8 `Standard` for bool: the most significant bit of ONE u32."""
    return rng.next_u32() >> 31 == 1


def _fq_sqrt(c, a: int):
    """q = 3 mod 4 for both base fields (ark-ff `SqrtPrecomputation::Case3Mod4`): a^((q + 1) / 4), None when a is not a square."""
    assert c.q % 4 == 3
    y = pow(a, (c.q + 1) // 4, c.q)
    return y if y * y % c.q == a % c.q else None


def g1_rand(c, rng: BlockRng):
    """`E::G1::rand(rng)` (snark.rs:496): affine point or None; get_point_from_x_unchecked orders the two roots as integers
    (`y < -y` on canonical values) and `greatest` picks the larger."""
    import pyref as P
    while True:
        x = fq_rand(c, rng)
        greatest = gen_bool(rng)
        y = _fq_sqrt(c, (x * x % c.q * x + c.b) % c.q)
        if y is None:
            continue
        small, large = sorted((y, (-y) % c.q))
        return P.g1_mul(c, G1_COFACTOR[c.curve_id], (x, large if greatest else small))


def g2_rand_consume(c, rng: BlockRng) -> int:
    """`E::G2::rand(rng)` (snark.rs:497) as far as the PROVER can tell: the draws it takes from the stream.  h enters the verifying
    key's open key only (not the transcript, not the proof), so the point itself is not rebuilt here.  Fq2 = Fq[u] / (u^2 + 1) on
    both curves; x^3 + b' (b' = 4 (1 + u) on BLS12-381, 3 / (9 + u) on BN254) has a square root iff its norm is a square in Fq
    (ark-ff QuadExtField::sqrt, complex method; c1 = 0 has probability 2^-254).  Returns the number of attempts."""
    q = c.q
    mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)
    if c.curve_id == 0:
        b2 = (4, 4)
    else:
        inv82 = pow(82, -1, q)
        b2 = (27 * inv82 % q, (-3) * inv82 % q)                                # 3 / (9 + u) = 3 (9 - u) / 82
    attempts = 0
    while True:
        attempts += 1
        x = (fq_rand(c, rng), fq_rand(c, rng))                                  # QuadExtField::rand: c0, then c1
        gen_bool(rng)
        x3 = mul(mul(x, x), x)
        rhs = ((x3[0] + b2[0]) % q, (x3[1] + b2[1]) % q)
        norm = (rhs[0] * rhs[0] + rhs[1] * rhs[1]) % q
        if rhs[1] == 0:
            raise NotImplementedError("c1 = 0: probability 2^-254")
        if norm == 0 or pow(norm, (q - 1) // 2, q) == 1:
            return attempts


def universal_setup_for_testing(c, rng: BlockRng):
    """(beta, g) of `universal_setup_for_testing` and the stream advanced past h: powers_of_g[i] = beta^i g (snark.rs:495-517)."""
    beta = fr_rand(c, rng)
    g = g1_rand(c, rng)
    g2_rand_consume(c, rng)
    return beta, g
