"""oracle/pyref_fs.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

The Fiat-Shamir side of `PlonkKzgSnark::prove`, restated from the published constructions, with NO import from the
product package (mpc-jellyfish_amd/transcript.py is a separate text by design: this file is the checker's copy):

  * Keccak-f[1600]            FIPS 202 section 3: the rotation offsets and the round constants are GENERATED here by the
                              rules of 3.2.2 (rho) and 3.2.5 (rc, the LFSR x^8 + x^6 + x^5 + x^4 + 1), not tabulated
  * Keccak-256 / SHA3-256     FIPS 202 sponge, rate 136, padding 0x01 / 0x06 ... 0x80
  * STROBE-128 (lite)         STROBE v1.0.2 section 6-7 as cut down by the `merlin` crate (src/strobe.rs: only AD, meta-AD, PRF, KEY)
  * merlin::Transcript        merlin 3.0 src/transcript.rs: "Merlin v1.0", dom-sep, LE32 length framing
  * StandardTranscript        /root/reference/plonk/src/transcript/standard.rs:16-46
  * PlonkTranscript order     /root/reference/plonk/src/transcript/mod.rs:45-214
  * SolidityTranscript        /root/reference/plonk/src/transcript/solidity.rs:31-78 (only its Keccak-256 KAT, :80-96, is used here)
  * ark-serialize 0.4         compressed G1 / Fr encodings behind `to_bytes!` (utilities/src/macros.rs:13-18) and
                              `Proof::serialize_compressed` (plonk/src/proof_system/structs.rs:59-84)

Pins (tests/test_oracle_fs.py): SHA3-256 against hashlib for many lengths; the reference's own Keccak-256 vector
(solidity.rs:80-96); merlin's published test transcript; the compressed BLS12-381 generator of the IETF / Zcash format.
"""
from __future__ import annotations

import struct

MASK64 = (1 << 64) - 1


# ---------------------------------------------------------------------------------------------------------------
# Keccak-f[1600], FIPS 202 section 3.2-3.3.  State = 25 lanes, lane (x, y) at index x + 5 y.
# ---------------------------------------------------------------------------------------------------------------
def _rho_offsets():
    """3.2.2: (x, y) starts at (1, 0); for t = 0..23 the offset of the current lane is (t+1)(t+2)/2 and (x, y) <- (y, 2x + 3y)."""
    off = [0] * 25
    x, y = 1, 0
    for t in range(24):
        off[x + 5 * y] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    return off


def _round_constants():
    """3.2.5: rc(t) is the output bit of the LFSR x^8 + x^6 + x^5 + x^4 + 1; RC[i] sets bit 2^j - 1 to rc(j + 7 i), j = 0..6."""
    def rc_bit(t):
        if t % 255 == 0:
            return 1
        reg = 1                                            # R = 10000000, R[0] is the low bit here
        for _ in range(t % 255):
            reg <<= 1
            if reg & 0x100:
                reg ^= 0x171                               # x^8 + x^6 + x^5 + x^4 + 1
        return reg & 1
    out = []
    for i in range(24):
        v = 0
        for j in range(7):
            if rc_bit(j + 7 * i):
                v |= 1 << ((1 << j) - 1)
        out.append(v)
    return out


_RHO = _rho_offsets()
_RC = _round_constants()


def _rot(v, n):
    return ((v << n) | (v >> (64 - n))) & MASK64 if n else v


def keccak_f1600(lanes):
    """24 rounds of theta, rho, pi, chi, iota on a list of 25 ints; returns the new list."""
    a = list(lanes)
    for rnd in range(24):
        col = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]                       # theta
        d = [col[(x + 4) % 5] ^ _rot(col[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25                                                                                        # rho + pi: (x, y) -> (y, 2x + 3y)
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = _rot(a[x + 5 * y], _RHO[x + 5 * y])
        a = [b[i] ^ (~b[(i % 5 + 1) % 5 + 5 * (i // 5)] & MASK64 & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) for i in range(25)]   # chi
        a[0] ^= _RC[rnd]                                                                                    # iota
    return a


def keccak_f1600_bytes(state: bytes) -> bytes:
    return struct.pack("<25Q", *keccak_f1600(struct.unpack("<25Q", state)))


def _sponge(msg: bytes, rate: int, suffix: int, out_len: int) -> bytes:
    buf = bytearray(msg) + bytes([suffix])
    buf += bytes(-len(buf) % rate)
    buf[-1] |= 0x80
    st = bytearray(200)
    for off in range(0, len(buf), rate):
        for i in range(rate):
            st[i] ^= buf[off + i]
        st = bytearray(keccak_f1600_bytes(bytes(st)))
    return bytes(st[:out_len])                              # out_len <= rate for the two hashes below


def keccak256(msg: bytes) -> bytes:
    """sha3::Keccak256 (the pre-standard padding 0x01), as used by SolidityTranscript."""
    return _sponge(msg, 136, 0x01, 32)


def sha3_256(msg: bytes) -> bytes:
    return _sponge(msg, 136, 0x06, 32)


# ---------------------------------------------------------------------------------------------------------------
# STROBE-128 as merlin cuts it (merlin src/strobe.rs)
# ---------------------------------------------------------------------------------------------------------------
STROBE_R = 166                                            # 200 - 2 * 16 (128-bit security) - 2
F_I, F_A, F_C, F_T, F_M, F_K = 1, 2, 4, 8, 16, 32


class Strobe:
    def __init__(self, protocol: bytes):
        init = bytes([1, STROBE_R + 2, 1, 0, 1, 96]) + b"STROBEv1.0.2"
        self.st = bytearray(keccak_f1600_bytes(init + bytes(200 - len(init))))
        self.pos = 0
        self.begin = 0
        self.flags = 0
        self.meta_ad(protocol, False)

    def _permute(self):
        self.st[self.pos] ^= self.begin
        self.st[self.pos + 1] ^= 0x04
        self.st[STROBE_R + 1] ^= 0x80
        self.st = bytearray(keccak_f1600_bytes(bytes(self.st)))
        self.pos = self.begin = 0

    def _absorb(self, data: bytes):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == STROBE_R:
                self._permute()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray()
        for _ in range(n):
            out.append(self.st[self.pos])
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._permute()
        return bytes(out)

    def _op(self, flags: int, more: bool):
        if more:
            if flags != self.flags:
                raise ValueError("continued operation with different flags")
            return
        if flags & F_T:
            raise ValueError("transport operations are not part of merlin's STROBE")
        prev = self.begin
        self.begin = self.pos + 1
        self.flags = flags
        self._absorb(bytes([prev, flags]))
        if flags & (F_C | F_K) and self.pos:
            self._permute()

    def meta_ad(self, data: bytes, more: bool):
        self._op(F_M | F_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._op(F_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool = False) -> bytes:
        self._op(F_I | F_A | F_C, more)
        return self._squeeze(n)


class Merlin:
    """merlin::Transcript (src/transcript.rs)."""

    def __init__(self, label: bytes):
        self.s = Strobe(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, msg: bytes):
        self.s.meta_ad(label, False)
        self.s.meta_ad(struct.pack("<I", len(msg)), True)
        self.s.ad(msg, False)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.s.meta_ad(label, False)
        self.s.meta_ad(struct.pack("<I", n), True)
        return self.s.prf(n)


# ---------------------------------------------------------------------------------------------------------------
# ark-serialize 0.4 compressed encodings.  `c` is a pyref.Curve.
# ---------------------------------------------------------------------------------------------------------------
def fr_bytes(c, x: int) -> bytes:
    """Fp<_, 4>::serialize_compressed: the canonical integer, 32 bytes little-endian (no flag bits for a prime field)."""
    return (x % c.r).to_bytes(32, "little")


def g1_bytes(c, pt) -> bytes:
    """Affine<G1>::serialize_compressed.  pt = (x, y) canonical integers, or None for the point at infinity.

    BN254 (ark-ec short-Weierstrass default, SWFlags): x little-endian in 32 bytes; bit 7 of the last byte = "y is the
    lexicographically larger root" (y > -y), bit 6 = infinity (then x = 0).
    BLS12-381 (ark-bls12-381 overrides it with the IETF / Zcash format): x big-endian in 48 bytes; bit 7 of the first byte =
    compressed, bit 6 = infinity, bit 5 = y is the larger root."""
    if c.curve_id == 0:
        if pt is None:
            return bytes([0xC0]) + bytes(47)
        x, y = pt
        head = bytearray(x.to_bytes(48, "big"))
        head[0] |= 0x80 | (0x20 if y > c.q - y else 0)
        return bytes(head)
    if pt is None:
        return bytes(31) + bytes([0x40])
    x, y = pt
    tail = bytearray(x.to_bytes(32, "little"))
    if y > c.q - y:
        tail[31] |= 0x80
    return bytes(tail)


def vec_bytes(items, enc) -> bytes:
    """Vec<T>::serialize_compressed: the length as u64 little-endian, then the items."""
    return struct.pack("<Q", len(items)) + b"".join(enc(i) for i in items)


# ---------------------------------------------------------------------------------------------------------------
# the two transcripts of the reference
# ---------------------------------------------------------------------------------------------------------------
class StandardTranscript:
    """plonk/src/transcript/standard.rs:16-46 with the default methods of `PlonkTranscript` (transcript/mod.rs:45-214).
    `usize` fields are 8 bytes (64-bit target), `MODULUS_BIT_SIZE` is a u32."""

    def __init__(self, c, label: bytes = b"PlonkProof"):
        self.c = c
        self.m = Merlin(label)

    def append_message(self, label: bytes, msg: bytes):
        self.m.append_message(label, msg)

    def append_field_elem(self, label: bytes, x: int):
        self.m.append_message(label, fr_bytes(self.c, x))

    def append_commitment(self, label: bytes, pt):
        self.m.append_message(label, g1_bytes(self.c, pt))

    def append_commitments(self, label: bytes, pts):
        for p in pts:
            self.append_commitment(label, p)

    def append_vk_and_pub_input(self, domain_size, num_inputs, k, selector_comms, sigma_comms, pub_input):
        """transcript/mod.rs:45-104"""
        self.append_message(b"field size in bits", struct.pack("<I", self.c.r.bit_length()))
        self.append_message(b"domain size", struct.pack("<Q", domain_size))
        self.append_message(b"input size", struct.pack("<Q", num_inputs))
        for ki in k:
            self.append_field_elem(b"wire subsets separators", ki)
        self.append_commitments(b"selector commitments", selector_comms)
        self.append_commitments(b"sigma commitments", sigma_comms)
        for x in pub_input:
            self.append_field_elem(b"public input", x)

    def append_plookup_evaluations(self, ev: dict):
        """transcript/mod.rs:165-202: six of the fifteen Plookup evaluations, under these labels, in this order."""
        for label, key in ((b"lookup_table_eval", "range_table_eval"), (b"h_1_eval", "h_1_eval"), (b"prod_next_eval", "prod_next_eval"),
                           (b"lookup_table_next_eval", "range_table_next_eval"), (b"h_1_next_eval", "h_1_next_eval"),
                           (b"h_2_next_eval", "h_2_next_eval")):
            self.append_field_elem(label, ev[key])

    def get_and_append_challenge(self, label: bytes) -> int:
        """standard.rs:33-45: 64 squeezed bytes, `from_le_bytes_mod_order`, and the challenge is absorbed again."""
        ch = int.from_bytes(self.m.challenge_bytes(label, 64), "little") % self.c.r
        self.append_field_elem(label, ch)
        return ch


class SolidityTranscript:
    """plonk/src/transcript/solidity.rs:31-78 (labels dropped; state = keccak256(state|transcript|0) || keccak256(..|1);
    challenge = the first 48 bytes of the state, little-endian, mod r).  Not used by the bench configs; kept because its
    hash is the one primitive of the path for which the reference holds a known-answer vector."""

    def __init__(self, c, _label: bytes = b""):
        self.c = c
        self.buf = b""
        self.state = bytes(64)

    def append_message(self, _label: bytes, msg: bytes):
        self.buf += msg

    def get_and_append_challenge(self, _label: bytes) -> int:
        self.state = keccak256(self.state + self.buf + b"\x00") + keccak256(self.state + self.buf + b"\x01")
        return int.from_bytes(self.state[:48], "little") % self.c.r
