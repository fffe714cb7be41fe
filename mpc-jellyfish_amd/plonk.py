"""TurboPlonk quotient round on the GPU -- mirror of the prover-side surface it replaces:

    ProvingKey{selectors, sigmas, vk.k}                plonk/src/proof_system/structs.rs:575-590
    Prover::compute_quotient_polynomial(challenges, pks, online_oracles, num_wire_types)
                                                        plonk/src/proof_system/prover.rs:512-673
    Oracles{wire_polys, pub_inp_poly, prod_perm_poly}   structs.rs:875-887
    Challenges{alpha, beta, gamma, ...}                 structs.rs:863-872

One instance, no Plookup.  Polynomials are (len,4) uint64 Montgomery coefficient arrays.  The proving
key's 18 fixed polynomials are transformed to the quotient coset once and stay in HBM.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import lib as _lib
from .params import CurveParams, curve as _curve, fr_to_mont

GATE_WIDTH = 4                 # relation/src/constants.rs:18
N_TURBO_PLONK_SELECTORS = 13   # relation/src/constants.rs:22
NUM_WIRE_TYPES = GATE_WIDTH + 1


@dataclass
class Challenges:
    """plonk/src/proof_system/structs.rs:863-872 (the three the quotient needs), Python ints."""
    alpha: int
    beta: int
    gamma: int


class PlonkError(Exception):
    """plonk/src/errors.rs:16-50."""


class ProvingKeyDevice:
    """selectors/sigmas of a ProvingKey, resident on the GPU as coset evaluations."""

    def __init__(self, curve: CurveParams, handle: int, domain_size: int):
        self.curve, self.handle, self.domain_size = curve, handle, domain_size

    @classmethod
    def register(cls, curve, domain_size: int, selectors, sigmas, k) -> "ProvingKeyDevice":
        """selectors: 13 coefficient arrays, sigmas: 5, k: 5 Python ints (coset representatives)."""
        c = _curve(curve)
        if domain_size & (domain_size - 1) or domain_size < 2:
            raise PlonkError("domain size must be a power of two")
        if len(selectors) != N_TURBO_PLONK_SELECTORS or len(sigmas) != NUM_WIRE_TYPES or len(k) != NUM_WIRE_TYPES:
            raise PlonkError("TurboPlonk proving key: 13 selectors, 5 sigmas, 5 coset representatives")
        polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in list(selectors) + list(sigmas)]
        plen = max(p.shape[0] for p in polys)
        slab = np.zeros((len(polys), plen, 4), dtype=np.uint64)
        for i, p in enumerate(polys):
            slab[i, :p.shape[0]] = p
        kk = fr_to_mont(c, list(k))
        L = _lib.ensure_init()
        h = C.c_uint64()
        _lib.check(L.mzk_plonk_pk_register(c.curve_id, domain_size.bit_length() - 1, NUM_WIRE_TYPES,
                                           slab[:N_TURBO_PLONK_SELECTORS].ctypes.data_as(C.c_void_p),
                                           np.ascontiguousarray(slab[N_TURBO_PLONK_SELECTORS:]).ctypes.data_as(C.c_void_p),
                                           plen, kk.ctypes.data_as(C.c_void_p), C.byref(h)), "mzk_plonk_pk_register")
        return cls(c, h.value, domain_size)

    def release(self):
        if self.handle:
            _lib.check(_lib.load().mzk_plonk_pk_release(self.handle), "mzk_plonk_pk_release")
            self.handle = 0


def compute_quotient_polynomial(pk: ProvingKeyDevice, challenges: Challenges, wire_polys, prod_perm_poly, pub_inp_poly) -> np.ndarray:
    """prover.rs:512-673 for one instance: returns the 8n coefficients of the quotient polynomial
    (callers strip trailing zeros as DensePolynomial::from_coefficients_vec does)."""
    if len(wire_polys) != NUM_WIRE_TYPES:
        raise PlonkError("inconsistent pks/online oracles when computing quotient polys")      # prover.rs:519-524
    polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in list(wire_polys) + [prod_perm_poly, pub_inp_poly]]
    plen = max(1, max(p.shape[0] for p in polys))
    m = 8 * pk.domain_size
    if plen > m:
        raise PlonkError("polynomial longer than the quotient domain")
    slab = np.zeros((len(polys), plen, 4), dtype=np.uint64)
    for i, p in enumerate(polys):
        slab[i, :p.shape[0]] = p
    ch = fr_to_mont(pk.curve, [challenges.alpha, challenges.beta, challenges.gamma])
    out = np.empty((m, 4), dtype=np.uint64)
    _lib.check(_lib.ensure_init().mzk_plonk_quotient(pk.handle, slab.ctypes.data_as(C.c_void_p), plen, ch[0].ctypes.data_as(C.c_void_p),
                                                     ch[1].ctypes.data_as(C.c_void_p), ch[2].ctypes.data_as(C.c_void_p),
                                                     out.ctypes.data_as(C.c_void_p)), "mzk_plonk_quotient")
    return out


def compute_quotient_polynomial_dev(pk: ProvingKeyDevice, challenges: Challenges, polys_dev, in_len: int, out_dev, stream=None):
    """Device-resident form: polys_dev is a (7, 8n, 4) int64 CUDA tensor holding the coefficients of the
    5 wire polynomials, z and the public-input polynomial in its first in_len rows (overwritten with the
    coset evaluations); out_dev (8n, 4) receives the quotient coefficients.  Asynchronous."""
    import torch
    m = 8 * pk.domain_size
    assert polys_dev.shape == (NUM_WIRE_TYPES + 2, m, 4) and out_dev.shape == (m, 4)
    assert polys_dev.is_cuda and polys_dev.is_contiguous() and out_dev.is_contiguous() and polys_dev.dtype == torch.int64
    st = torch.cuda.current_stream(polys_dev.device).cuda_stream if stream is None else stream
    ch = fr_to_mont(pk.curve, [challenges.alpha, challenges.beta, challenges.gamma])
    _lib.check(_lib.ensure_init().mzk_plonk_quotient_dev(pk.handle, polys_dev.data_ptr(), in_len, ch[0].ctypes.data_as(C.c_void_p),
                                                         ch[1].ctypes.data_as(C.c_void_p), ch[2].ctypes.data_as(C.c_void_p),
                                                         out_dev.data_ptr(), st), "mzk_plonk_quotient_dev")
    return out_dev


def compute_prod_permutation_polynomial(pk: ProvingKeyDevice, beta: int, gamma: int, wire_values) -> np.ndarray:
    """relation/src/constraint_system.rs:1197-1223: the permutation grand product z, as n coefficients.
    wire_values: (5, n, 4) Montgomery wire evaluations (witness[wire_variable(i, j)])."""
    w = np.ascontiguousarray(wire_values, dtype=np.uint64)
    if w.shape != (NUM_WIRE_TYPES, pk.domain_size, 4):
        raise PlonkError("expected (5, n, 4) wire values")
    ch = fr_to_mont(pk.curve, [beta, gamma])
    out = np.empty((pk.domain_size, 4), dtype=np.uint64)
    _lib.check(_lib.ensure_init().mzk_plonk_perm_product(pk.handle, w.ctypes.data_as(C.c_void_p), ch[0].ctypes.data_as(C.c_void_p),
                                                         ch[1].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "mzk_plonk_perm_product")
    return out
