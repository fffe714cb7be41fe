"""Curve and field parameters plus host-side encodings (values as in SURVEY.md Appendix A).
Field elements cross the boundary as little-endian u64 limbs in Montgomery form (R = 2^(64*limbs)),
the in-memory image of ark-ff's Fp<MontBackend<_,N>,N>."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class CurveParams:
    name: str
    curve_id: int
    r: int
    q: int
    fr_generator: int       # Fr::GENERATOR, the coset offset of plonk/src/proof_system/prover.rs:545
    two_adicity: int
    b: int
    gx: int
    gy: int
    fq_limbs: int

    @property
    def fr_limbs(self) -> int:
        return 4

    @property
    def fr_R(self) -> int:
        return 1 << 256

    @property
    def fq_R(self) -> int:
        return 1 << (64 * self.fq_limbs)

    @property
    def fr_Rinv(self) -> int:
        return _inverse_of_R(self.r, 256)

    @property
    def fq_Rinv(self) -> int:
        return _inverse_of_R(self.q, 64 * self.fq_limbs)


_RINV: dict = {}


def _inverse_of_R(p: int, bits: int) -> int:
    if (p, bits) not in _RINV:
        _RINV[(p, bits)] = pow(1 << bits, -1, p)
    return _RINV[(p, bits)]


BLS12_381 = CurveParams(
    "bls12-381", 0,
    0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    7, 32, 4,
    0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
    6)
BN254 = CurveParams(
    "bn254", 1,
    21888242871839275222246405745257275088548364400416034343698204186575808495617,
    21888242871839275222246405745257275088696311157297823662689037894645226208583,
    5, 28, 3, 1, 2, 4)
CURVES = {0: BLS12_381, 1: BN254, "bls12-381": BLS12_381, "bn254": BN254}


def curve(c) -> CurveParams:
    return c if isinstance(c, CurveParams) else CURVES[c]


def int_to_limbs(x: int, n_limbs: int) -> np.ndarray:
    return np.frombuffer(x.to_bytes(8 * n_limbs, "little"), dtype="<u8").copy()


def limbs_to_int(a) -> int:
    return int.from_bytes(np.ascontiguousarray(a, dtype="<u8").tobytes(), "little")


def _ints_to_limbs(values, n_limbs: int) -> np.ndarray:
    """little-endian 64-bit limbs of each value, through one bytes object (these conversions sit between the kernels of a proof)"""
    nb = 8 * n_limbs
    return np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in values), dtype="<u8").reshape(-1, n_limbs).copy()


def _limbs_to_ints(a, n_limbs: int) -> list[int]:
    raw = np.ascontiguousarray(a, dtype="<u8").reshape(-1, n_limbs).tobytes()
    nb = 8 * n_limbs
    return [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]


def fr_to_mont(c: CurveParams, values) -> np.ndarray:
    """Python ints (canonical) -> (n,4) uint64 Montgomery."""
    r, R = c.r, c.fr_R
    return _ints_to_limbs([int(v) % r * R % r for v in values], 4) if len(values) else np.zeros((0, 4), np.uint64)


def fr_from_mont(c: CurveParams, a: np.ndarray) -> list[int]:
    rinv, r = c.fr_Rinv, c.r
    return [v * rinv % r for v in _limbs_to_ints(a, 4)]


def fr_bigints(values) -> np.ndarray:
    """Python ints -> (n,4) uint64 canonical integers (msm_bigint scalars)."""
    return _ints_to_limbs([int(v) for v in values], 4) if len(values) else np.zeros((0, 4), np.uint64)


def fq_to_mont(c: CurveParams, values) -> np.ndarray:
    q, R = c.q, c.fq_R
    return _ints_to_limbs([int(v) % q * R % q for v in values], c.fq_limbs)


def fq_from_mont(c: CurveParams, a: np.ndarray) -> list[int]:
    rinv, q = c.fq_Rinv, c.q
    return [v * rinv % q for v in _limbs_to_ints(a, c.fq_limbs)]


def random_fr_mont(c: CurveParams, n: int, seed: int) -> np.ndarray:
    """n uniformly random field elements as (n,4) uint64 limbs (< r, hence valid in either form),
    by 256-bit rejection sampling with a seeded generator."""
    rng = np.random.default_rng(seed)
    bits = c.r.bit_length()
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    mod = int_to_limbs(c.r, 4)
    out = np.empty((n, 4), dtype=np.uint64)
    filled = 0
    while filled < n:
        m = max(1024, int((n - filled) * 2.3))
        cand = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64)
        cand[:, 3] &= top_mask
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for j in (3, 2, 1, 0):
            lt |= eq & (cand[:, j] < mod[j])
            eq &= cand[:, j] == mod[j]
        good = cand[lt]
        k = min(len(good), n - filled)
        out[filled:filled + k] = good[:k]
        filled += k
    return out


def domain_size_ratio(n: int, num_wire_types: int) -> int:
    """plonk/src/constants.rs:18-20."""
    return (num_wire_types * (n + 1) + 2) // n + 1
