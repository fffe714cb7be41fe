"""TurboPlonk / UltraPlonk quotient round and grand products on the GPU -- mirror of the prover-side surface they replace:

    ProvingKey{selectors, sigmas, vk.k}                plonk/src/proof_system/structs.rs:575-590
    Prover::compute_quotient_polynomial(challenges, pks, online_oracles, num_wire_types)
                                                        plonk/src/proof_system/prover.rs:512-673
    Oracles{wire_polys, pub_inp_poly, prod_perm_poly}   structs.rs:875-887
    Challenges{alpha, beta, gamma, ...}                 structs.rs:863-872

    PlookupProvingKey{range_table_poly, key_table_poly, table_dom_sep_poly, q_dom_sep_poly}   structs.rs:592-640
    Arithmetization::compute_lookup_sorted_vec_polynomials / compute_lookup_prod_polynomial    relation/src/constraint_system.rs:1311-1417

One instance.  Polynomials are (len,4) uint64 Montgomery coefficient arrays.  The proving key's 18 (UltraPlonk: 24)
fixed polynomials are transformed to the quotient coset once and stay in HBM.
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
N_ULTRA_PLONK_SELECTORS = 14   # + q_lookup
NUM_WIRE_TYPES_ULTRA = GATE_WIDTH + 2
PLOOKUP_TABLE_POLYS = ("range_table_poly", "key_table_poly", "table_dom_sep_poly", "q_dom_sep_poly")


@dataclass
class Challenges:
    """plonk/src/proof_system/structs.rs:863-872 (the three the quotient needs), Python ints."""
    alpha: int
    beta: int
    gamma: int
    tau: int = 0


class PlonkError(Exception):
    """plonk/src/errors.rs:16-50; `kind` names the variant when the reference has a dedicated one (e.g. WrongQuotientPolyDegree)."""

    def __init__(self, msg: str, kind: str | None = None):
        super().__init__(msg if kind is None else "%s: %s" % (kind, msg))
        self.kind = kind


class ProvingKeyDevice:
    """selectors/sigmas (and Plookup table polynomials) of a ProvingKey, resident on the GPU as coset evaluations."""

    def __init__(self, curve: CurveParams, handle: int, domain_size: int, ultra: bool = False, classes=None):
        self.curve, self.handle, self.domain_size, self.ultra = curve, handle, domain_size, ultra
        self.num_wire_types = NUM_WIRE_TYPES_ULTRA if ultra else NUM_WIRE_TYPES
        self.classes = None if classes is None else list(classes)      # residue classes mod 8 resident here (coset-chunked key)

    @classmethod
    def register(cls, curve, domain_size: int, selectors, sigmas, k, plookup=None, classes=None) -> "ProvingKeyDevice":
        """selectors: 13 coefficient arrays, sigmas: 5, k: 5 Python ints (coset representatives).
        UltraPlonk: 14 selectors (q_lookup last), 6 sigmas, 6 k and plookup = {range_table_poly, key_table_poly,
        table_dom_sep_poly, q_dom_sep_poly} coefficient arrays.
        classes: None for the whole quotient domain, or the residue classes mod 8 (strictly increasing) this GPU keeps
        -- the coset-chunked key of SURVEY.md 8(e).3, used with compute_quotient_chunked_dev."""
        c = _curve(curve)
        ultra = plookup is not None
        nsel, W = (N_ULTRA_PLONK_SELECTORS, NUM_WIRE_TYPES_ULTRA) if ultra else (N_TURBO_PLONK_SELECTORS, NUM_WIRE_TYPES)
        if domain_size & (domain_size - 1) or domain_size < 2:
            raise PlonkError("domain size must be a power of two")
        if len(selectors) != nsel or len(sigmas) != W or len(k) != W:
            raise PlonkError("proving key: %d selectors, %d sigmas, %d coset representatives" % (nsel, W, W))
        tabs = [plookup[x] for x in PLOOKUP_TABLE_POLYS] if ultra else []
        polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in list(selectors) + list(sigmas) + tabs]
        plen = max(p.shape[0] for p in polys)
        slab = np.zeros((len(polys), plen, 4), dtype=np.uint64)
        for i, p in enumerate(polys):
            slab[i, :p.shape[0]] = p
        kk = fr_to_mont(c, list(k))
        L = _lib.ensure_init()
        h = C.c_uint64()
        ptr = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        sel_a, sig_a = np.ascontiguousarray(slab[:nsel]), np.ascontiguousarray(slab[nsel:nsel + W])
        if classes is not None:
            tab_a = np.ascontiguousarray(slab[nsel + W:]) if ultra else None
            ca = np.ascontiguousarray(list(classes), dtype=np.uint32)
            _lib.check(L.mzk_plonk_pk_register_chunked(c.curve_id, domain_size.bit_length() - 1, W, ptr(sel_a), ptr(sig_a), ptr(tab_a) if ultra else None, plen,
                                                       ptr(kk), C.c_void_p(ca.ctypes.data), len(ca), C.byref(h)), "mzk_plonk_pk_register_chunked")
        elif ultra:
            tab_a = np.ascontiguousarray(slab[nsel + W:])
            _lib.check(L.mzk_plonk_pk_register_ultra(c.curve_id, domain_size.bit_length() - 1, ptr(sel_a), ptr(sig_a), ptr(tab_a), plen, ptr(kk), C.byref(h)),
                       "mzk_plonk_pk_register_ultra")
        else:
            _lib.check(L.mzk_plonk_pk_register(c.curve_id, domain_size.bit_length() - 1, W, ptr(sel_a), ptr(sig_a), plen, ptr(kk), C.byref(h)),
                       "mzk_plonk_pk_register")
        return cls(c, h.value, domain_size, ultra, classes)

    def release(self):
        if self.handle:
            _lib.check(_lib.load().mzk_plonk_pk_release(self.handle), "mzk_plonk_pk_release")
            self.handle = 0


def compute_quotient_polynomial(pk: ProvingKeyDevice, challenges: Challenges, wire_polys, prod_perm_poly, pub_inp_poly) -> np.ndarray:
    """prover.rs:512-673 for one instance: returns the 8n coefficients of the quotient polynomial
    (callers strip trailing zeros as DensePolynomial::from_coefficients_vec does)."""
    if len(wire_polys) != NUM_WIRE_TYPES or pk.ultra:
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
    _lib.check(_lib.ensure_init().mzk_plonk_quotient(pk.handle, C.c_void_p(slab.ctypes.data), plen, C.c_void_p(ch[0].ctypes.data),
                                                     C.c_void_p(ch[1].ctypes.data), C.c_void_p(ch[2].ctypes.data),
                                                     C.c_void_p(out.ctypes.data)), "mzk_plonk_quotient")
    return out


def compute_quotient_polynomial_dev(pk: ProvingKeyDevice, challenges: Challenges, polys_dev, in_len: int, out_dev, stream=None):
    """Device-resident form: polys_dev is a (W + 2, 8n, 4) int64 CUDA tensor holding the coefficients of the
    wire polynomials, z and the public-input polynomial in its first in_len rows (overwritten with the
    coset evaluations) -- UltraPlonk: (6 + 2 + 3, 8n, 4) with h_1, h_2 and the Plookup product polynomial appended;
    out_dev (8n, 4) receives the quotient coefficients.  Asynchronous."""
    import torch
    m = 8 * pk.domain_size
    rows = pk.num_wire_types + 2 + (3 if pk.ultra else 0)
    assert polys_dev.shape == (rows, m, 4) and out_dev.shape == (m, 4)
    assert polys_dev.is_cuda and polys_dev.is_contiguous() and out_dev.is_contiguous() and polys_dev.dtype == torch.int64
    st = torch.cuda.current_stream(polys_dev.device).cuda_stream if stream is None else stream
    ch = fr_to_mont(pk.curve, [challenges.alpha, challenges.beta, challenges.gamma, challenges.tau])
    p = lambda i: C.c_void_p(ch[i].ctypes.data)
    if pk.ultra:
        _lib.check(_lib.ensure_init().mzk_plonk_quotient_ultra_dev(pk.handle, polys_dev.data_ptr(), in_len, p(3), p(0), p(1), p(2), out_dev.data_ptr(), st),
                   "mzk_plonk_quotient_ultra_dev")
    else:
        _lib.check(_lib.ensure_init().mzk_plonk_quotient_dev(pk.handle, polys_dev.data_ptr(), in_len, p(0), p(1), p(2), out_dev.data_ptr(), st),
                   "mzk_plonk_quotient_dev")
    return out_dev


def compute_quotient_chunked_dev(pk: ProvingKeyDevice, challenges: Challenges, polys_dev, in_len: int, out_dev=None, stream=None, pi_zero: bool = False):
    """SURVEY.md 8(e).3, local part: polys_dev is a (W + 2 [+ 3], stride, 4) CUDA tensor of coefficient rows (first in_len
    <= 2n slots used, not overwritten); returns (len(pk.classes), n, 4): per resident class k the coefficients of
    t mod (X^n - h_k^n).  pi_zero: the public-input polynomial (row W + 1) is known to be zero -- it is then neither transformed nor
    read (MZK_QUOTIENT_PI_ZERO).  Asynchronous."""
    import torch
    n = pk.domain_size
    rows = pk.num_wire_types + 2 + (3 if pk.ultra else 0)
    assert pk.classes is not None and polys_dev.dim() == 3 and polys_dev.shape[0] == rows and polys_dev.shape[2] == 4
    assert polys_dev.is_cuda and polys_dev.is_contiguous() and polys_dev.dtype == torch.int64
    out = torch.empty((len(pk.classes), n, 4), dtype=torch.int64, device=polys_dev.device) if out_dev is None else out_dev
    st = torch.cuda.current_stream(polys_dev.device).cuda_stream if stream is None else stream
    ch = fr_to_mont(pk.curve, [challenges.alpha, challenges.beta, challenges.gamma, challenges.tau])
    p = lambda i: C.c_void_p(ch[i].ctypes.data)
    _lib.check(_lib.ensure_init().mzk_plonk_quotient_chunked_flags_dev(pk.handle, polys_dev.data_ptr(), polys_dev.shape[1], in_len, 1 if pi_zero else 0,
                                                                       p(3) if pk.ultra else None, p(0), p(1), p(2), out.data_ptr(), st),
               "mzk_plonk_quotient_chunked_flags_dev")
    return out


def quotient_top_supported(num_wire_types: int, domain_size: int) -> bool:
    """n > W + 2 (and n >= 8): the coefficients of the quotient from X^(Wn) on are then the top W + 3 coefficients of its numerator
    (include/mzk.h, mzk_plonk_quotient_top_dev)."""
    return domain_size > num_wire_types + 2 and domain_size >= 8 and num_wire_types <= 6       # (plonk_quotient_top_kernel: series slots for W <= 6)


def quotient_classes_needed(num_wire_types: int, domain_size: int, top: bool = True) -> list[int]:
    """The residue classes of the quotient domain that have to be evaluated.  deg t = W (n + 1) + 2 (prover.rs:916-919).
    top=True (the device provers): the W + 3 coefficients from X^(Wn) on come from compute_quotient_top_dev, so W classes determine
    the rest -- 5 of the 8 for TurboPlonk, 6 for UltraPlonk (n > W + 2; smaller domains fall back to the rule below).  The recovered
    polynomial then has the expected degree for ANY witness: the caller checks the quotient identity at zeta instead
    (prover.py, check_quotient_identity).
    top=False (host-pointer mzk_plonk_quotient, whose caller keeps the reference's degree guard): deg t is below (W + 1) n as soon as
    n > W + 2, so W + 1 classes determine it; tiny domains keep all 8.  One SPARE coefficient is required above the expected degree
    (n >= W + 4): the interpolant through (W + 1) n points has degree < (W + 1) n whatever the witness, so at n = W + 3, where that
    bound IS the expected degree, an unsatisfied witness would pass `WrongQuotientPolyDegree` (prover.rs:915-918)."""
    W, n = num_wire_types, domain_size
    if top and quotient_top_supported(W, n):
        return list(range(W))
    return list(range(W + 1)) if W * (n + 1) + 2 < (W + 1) * n - 1 and W + 1 <= 8 else list(range(8))


def compute_quotient_top_dev(pk: ProvingKeyDevice, challenges: Challenges, polys_dev, in_len: int, out_dev=None, stream=None):
    """The W + 3 coefficients of the quotient from X^(Wn) on, from the top coefficients of the numerator's factors (no evaluation):
    polys_dev as for compute_quotient_chunked_dev (rows: W wire polynomials, then z).  Returns a (16, 4) CUDA tensor, the first W + 3
    rows filled.  Asynchronous."""
    import torch
    assert pk.classes is not None and polys_dev.dim() == 3 and polys_dev.shape[2] == 4 and polys_dev.is_cuda and polys_dev.is_contiguous()
    out = torch.zeros((16, 4), dtype=torch.int64, device=polys_dev.device) if out_dev is None else out_dev
    st = torch.cuda.current_stream(polys_dev.device).cuda_stream if stream is None else stream
    ch = fr_to_mont(pk.curve, [challenges.alpha, challenges.beta, challenges.gamma])
    p = lambda i: C.c_void_p(ch[i].ctypes.data)
    _lib.check(_lib.ensure_init().mzk_plonk_quotient_top_dev(pk.handle, polys_dev.data_ptr(), polys_dev.shape[1], in_len, p(0), p(1), p(2), out.data_ptr(), None, st),
               "mzk_plonk_quotient_top_dev")
    return out


def combine_quotient_classes(curve, domain_size: int, class_remainders, classes=None, out_dev=None, stream=None, top=None, n_top: int = 0):
    """SURVEY.md 8(e).3, after the exchange: (len(classes), n, 4) class remainders (class-major, in the order of `classes`;
    default all 8) -> (8n, 4) quotient coefficients, what `coset.ifft` returns at prover.rs:672 (slabs above len(classes) zero).
    top / n_top: the result of compute_quotient_top_dev and W + 3 -- the classes then only have to determine the n len(classes)
    coefficients below.  Asynchronous."""
    import torch
    c = _curve(curve)
    n = domain_size
    r = class_remainders
    cl = list(range(8)) if classes is None else list(classes)
    assert tuple(r.shape) == (len(cl), n, 4) and r.is_cuda and r.is_contiguous()
    out = torch.empty((8 * n, 4), dtype=torch.int64, device=r.device) if out_dev is None else out_dev
    st = torch.cuda.current_stream(r.device).cuda_stream if stream is None else stream
    ca = np.ascontiguousarray(cl, dtype=np.uint32)
    L = _lib.ensure_init()
    if top is None:
        _lib.check(L.mzk_plonk_quotient_combine_classes_dev(c.curve_id, n.bit_length() - 1, C.c_void_p(ca.ctypes.data), len(cl), r.data_ptr(), out.data_ptr(), st),
                   "mzk_plonk_quotient_combine_classes_dev")
    else:
        _lib.check(L.mzk_plonk_quotient_combine_top_dev(c.curve_id, n.bit_length() - 1, C.c_void_p(ca.ctypes.data), len(cl), r.data_ptr(), top.data_ptr(), n_top,
                                                        out.data_ptr(), st), "mzk_plonk_quotient_combine_top_dev")
    return out


def _to_dev(x):
    import torch
    if hasattr(x, "is_cuda"):
        return x.contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint64).view(np.int64)).cuda()


def compute_prod_permutation_polynomial_dev(pk: ProvingKeyDevice, beta: int, gamma: int, wire_values_dev, out_dev=None):
    """constraint_system.rs:1197-1223 on device tensors: (W, n, 4) wire values -> (n, 4) coefficients.  Asynchronous."""
    import torch
    n = pk.domain_size
    assert wire_values_dev.shape == (pk.num_wire_types, n, 4) and wire_values_dev.is_contiguous()
    out = torch.empty((n, 4), dtype=torch.int64, device=wire_values_dev.device) if out_dev is None else out_dev
    ch = fr_to_mont(pk.curve, [beta, gamma])
    _lib.check(_lib.ensure_init().mzk_plonk_perm_product_dev(pk.handle, wire_values_dev.data_ptr(), C.c_void_p(ch[0].ctypes.data),
                                                             C.c_void_p(ch[1].ctypes.data), out.data_ptr(),
                                                             torch.cuda.current_stream(wire_values_dev.device).cuda_stream), "mzk_plonk_perm_product_dev")
    return out


def compute_lookup_sorted_vec(pk: ProvingKeyDevice, tau: int, wire_values):
    """compute_merged_lookup_table + the merge of compute_lookup_sorted_vec_polynomials
    (constraint_system.rs:1290-1309, 1370-1408).  wire_values: (6, n, 4) numpy or CUDA tensor.
    Returns CUDA tensors (merged_table (n,4), merged_lookup_witness (n,4), sorted_vec (2n-1,4)); raises PlonkError
    like the reference when a lookup value is not in the table."""
    import torch
    if not pk.ultra:
        raise PlonkError("Mismatched Plonk types between the proving key and the circuit")       # snark.rs:249-254
    n = pk.domain_size
    w = _to_dev(wire_values)
    if tuple(w.shape) != (NUM_WIRE_TYPES_ULTRA, n, 4):
        raise PlonkError("expected (6, n, 4) wire values")
    table = torch.empty((n, 4), dtype=torch.int64, device=w.device)
    lookup = torch.empty((n, 4), dtype=torch.int64, device=w.device)
    sorted_vec = torch.empty((2 * n - 1, 4), dtype=torch.int64, device=w.device)
    t = fr_to_mont(pk.curve, [tau])
    rc = _lib.ensure_init().mzk_plookup_sorted_vec_dev(pk.handle, w.data_ptr(), C.c_void_p(t[0].ctypes.data), table.data_ptr(), lookup.data_ptr(),
                                                       sorted_vec.data_ptr(), torch.cuda.current_stream(w.device).cuda_stream)
    if rc == -8:
        raise PlonkError("The sorted vector has wrong length, some lookup variables might be outside the table")
    _lib.check(rc, "mzk_plookup_sorted_vec_dev")
    return table, lookup, sorted_vec


def compute_lookup_prod_polynomial(pk: ProvingKeyDevice, beta: int, gamma: int, merged_table, merged_lookup, sorted_vec, out_dev=None):
    """constraint_system.rs:1311-1368: coefficients (n, 4) of the Plookup product polynomial, CUDA tensor.  Asynchronous."""
    import torch
    n = pk.domain_size
    t, l, s = _to_dev(merged_table), _to_dev(merged_lookup), _to_dev(sorted_vec)
    if tuple(t.shape) != (n, 4) or tuple(l.shape) != (n, 4):
        raise PlonkError("Domain size should match the size of the padded lookup table")           # constraint_system.rs:1329-1332
    if tuple(s.shape) != (2 * n - 1, 4):
        raise PlonkError("The sorted vector has wrong length")                                      # constraint_system.rs:1334-1336
    out = torch.empty((n, 4), dtype=torch.int64, device=t.device) if out_dev is None else out_dev
    ch = fr_to_mont(pk.curve, [beta, gamma])
    _lib.check(_lib.ensure_init().mzk_plookup_product_dev(pk.handle, t.data_ptr(), l.data_ptr(), s.data_ptr(), C.c_void_p(ch[0].ctypes.data),
                                                          C.c_void_p(ch[1].ctypes.data), out.data_ptr(), torch.cuda.current_stream(t.device).cuda_stream),
               "mzk_plookup_product_dev")
    return out


def compute_prod_permutation_polynomial(pk: ProvingKeyDevice, beta: int, gamma: int, wire_values) -> np.ndarray:
    """relation/src/constraint_system.rs:1197-1223: the permutation grand product z, as n coefficients.
    wire_values: (W, n, 4) Montgomery wire evaluations (witness[wire_variable(i, j)])."""
    w = np.ascontiguousarray(wire_values, dtype=np.uint64)
    if w.shape != (pk.num_wire_types, pk.domain_size, 4):
        raise PlonkError("expected (%d, n, 4) wire values" % pk.num_wire_types)
    ch = fr_to_mont(pk.curve, [beta, gamma])
    out = np.empty((pk.domain_size, 4), dtype=np.uint64)
    _lib.check(_lib.ensure_init().mzk_plonk_perm_product(pk.handle, C.c_void_p(w.ctypes.data), C.c_void_p(ch[0].ctypes.data),
                                                         C.c_void_p(ch[1].ctypes.data), C.c_void_p(out.ctypes.data)), "mzk_plonk_perm_product")
    return out
