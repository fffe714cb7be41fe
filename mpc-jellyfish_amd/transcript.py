"""Fiat-Shamir transcript of the jf-plonk prover -- mirror of `StandardTranscript`
(plonk/src/transcript/standard.rs:16-46), a wrapper of merlin::Transcript (crate `merlin ^3.0`,
plonk/Cargo.toml; not vendored under the reference tree), with the message order of
`PlonkTranscript` (plonk/src/transcript/mod.rs:40-214) and `batch_prove_internal` (snark.rs:263-431).

Pure host code (the transcript hashes ~40 short messages per proof).  Merlin is
restated from its published construction: STROBE-128 (Strobe128 lite: R = 166, Keccak-f[1600]) under
the framing  meta-AD(label) || meta-AD(LE32(len)) || AD(message)  /  PRF(len).  Keccak-f[1600] is pinned
against hashlib's SHA3; the Merlin layer against the crate's published test transcript
("test protocol" / "some label" / "some data" -> d5a21972...9bca...0615, tests/test_transcript.py).  Serialisations follow ark-serialize
0.4 compressed forms: Fr = 32 bytes little-endian; BLS12-381 G1 = 48 bytes big-endian with the
compression / infinity / sign flags in the top bits (the IETF-Zcash form); BN254 G1 = 32 bytes
little-endian x with (y-is-negative, infinity) in the top bits of the last byte.
"""
from __future__ import annotations

from .params import CurveParams, curve as _curve

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600_py(state: bytearray) -> None:
    """In-place Keccak-f[1600] on 200 bytes (lane (x, y) at byte offset 8 (x + 5 y), little-endian); pure Python,
    kept as the cross-check of the library's host function."""
    a = [[int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


def keccak_f1600(state: bytearray) -> None:
    """In-place Keccak-f[1600] through the library's host-only entry point (no GPU involved)."""
    import ctypes as C
    from . import lib as _lib
    buf = (C.c_uint8 * 200).from_buffer(state)
    _lib.check(_lib.load().mzk_keccak_f1600(C.addressof(buf)), "mzk_keccak_f1600")


class Strobe128:
    R = 166
    FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32

    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[self.R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = self.pos_begin = 0

    def _absorb(self, data: bytes):
        for byte in data:
            self.state[self.pos] ^= byte
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray(n)
        for i in range(n):
            out[i] = self.state[self.pos]
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            assert self.cur_flags == flags
            return
        assert not flags & self.FLAG_T
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (self.FLAG_C | self.FLAG_K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_M | self.FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool = False) -> bytes:
        self._begin_op(self.FLAG_I | self.FLAG_A | self.FLAG_C, more)
        return self._squeeze(n)


class MerlinTranscript:
    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, message: bytes):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n)


# ---- ark-serialize 0.4 compressed encodings ---------------------------------------------------------------
def fr_bytes(c: CurveParams, x: int) -> bytes:
    return (x % c.r).to_bytes(32, "little")


def g1_bytes(c: CurveParams, point) -> bytes:
    """point: (x, y) canonical ints or None for infinity."""
    if c.curve_id == 0:                                   # BLS12-381: 48 bytes big-endian, flags in the top 3 bits
        if point is None:
            return bytes([0xC0]) + bytes(47)
        x, y = point
        b = bytearray(x.to_bytes(48, "big"))
        b[0] |= 0x80
        if y > (c.q - y) % c.q:
            b[0] |= 0x20
        return bytes(b)
    if point is None:                                     # arkworks short-Weierstrass default
        b = bytearray(32)
        b[31] |= 0x40
        return bytes(b)
    x, y = point
    b = bytearray(x.to_bytes(32, "little"))
    if y > (c.q - y) % c.q:
        b[31] |= 0x80
    return bytes(b)


class StandardTranscript:
    """plonk/src/transcript/standard.rs: Merlin with 64-byte challenges reduced mod r and re-absorbed."""

    def __init__(self, curve, label: bytes = b"PlonkProof"):
        self.curve = _curve(curve)
        self.t = MerlinTranscript(label)

    def append_message(self, label: bytes, msg: bytes):
        self.t.append_message(label, msg)

    def append_field_elem(self, label: bytes, x: int):
        self.append_message(label, fr_bytes(self.curve, x))

    def append_commitment(self, label: bytes, point):
        self.append_message(label, g1_bytes(self.curve, point))

    def append_commitments(self, label: bytes, points):
        for p in points:
            self.append_commitment(label, p)

    def append_vk_and_pub_input(self, domain_size: int, num_inputs: int, k, selector_comms, sigma_comms, pub_input):
        """transcript/mod.rs:45-104 (usize fields as 8-byte little-endian on a 64-bit target)."""
        c = self.curve
        self.append_message(b"field size in bits", c.r.bit_length().to_bytes(4, "little"))
        self.append_message(b"domain size", domain_size.to_bytes(8, "little"))
        self.append_message(b"input size", num_inputs.to_bytes(8, "little"))
        for ki in k:
            self.append_field_elem(b"wire subsets separators", ki)
        self.append_commitments(b"selector commitments", selector_comms)
        self.append_commitments(b"sigma commitments", sigma_comms)
        for x in pub_input:
            self.append_field_elem(b"public input", x)

    def append_plookup_evaluations(self, evals: dict) -> None:
        """transcript/mod.rs:165-202: six of the fifteen Plookup evaluations are absorbed."""
        for label, key in ((b"lookup_table_eval", "range_table_eval"), (b"h_1_eval", "h_1_eval"), (b"prod_next_eval", "prod_next_eval"),
                           (b"lookup_table_next_eval", "range_table_next_eval"), (b"h_1_next_eval", "h_1_next_eval"), (b"h_2_next_eval", "h_2_next_eval")):
            self.append_field_elem(label, evals[key])

    def get_and_append_challenge(self, label: bytes) -> int:
        buf = self.t.challenge_bytes(label, 64)
        ch = int.from_bytes(buf, "little") % self.curve.r
        self.append_field_elem(label, ch)
        return ch
