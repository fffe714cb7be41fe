"""Dense-polynomial primitives of prover rounds 4 and 5 on device-resident coefficient vectors --
mirror of the ark-poly `DensePolynomial` operations the jf-plonk prover uses there:

    poly.evaluate(&zeta)                          plonk/src/proof_system/prover.rs:216-235
    mul_poly(&p, &c), p + q                       prover.rs:302-358, 1115-1122, 497-501
    &batch_poly / &(X - z)                        prover.rs:504-506

Coefficient vectors are (len, 4) int64 CUDA tensors (Montgomery limbs); scalars are Python ints.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .params import curve as _curve, fr_from_mont, fr_to_mont


MAX_TERMS = 32                                              # POLY_MAX_TERMS of csrc/poly.cuh: terms of one lincomb launch


def _stream(t, stream):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream


def evaluate(curve, polys_dev, x: int, length: int | None = None, offset: int = 0) -> list[int]:
    """polys_dev: (len, 4) or (batch, stride, 4) CUDA tensor; every polynomial is evaluated at x over its
    first `length` coefficients.  offset: evaluate sum_{j < length} c[offset + j] x^j instead -- the coefficient range
    [offset, offset + length) read as a polynomial of its own (a rank's share of an evaluation: times x^offset it is that
    range's contribution, sharding.py).  Returns canonical Python ints (the call synchronises)."""
    c = _curve(curve)
    t = polys_dev
    if t.dim() == 2:
        batch, stride = 1, t.shape[0]
    else:
        batch, stride = t.shape[0], t.shape[1]
    n = stride - offset if length is None else length
    assert t.is_cuda and t.is_contiguous() and t.shape[-1] == 4 and 0 <= offset and n >= 0 and offset + n <= stride
    xm = fr_to_mont(c, [x])[0]
    out = np.empty((batch, 4), dtype=np.uint64)
    _lib.check(_lib.ensure_init().mzk_poly_eval_dev(c.curve_id, t.data_ptr() + 32 * offset, n, batch, stride, C.c_void_p(xm.ctypes.data),
                                                    C.c_void_p(out.ctypes.data), _stream(t, None)), "mzk_poly_eval_dev")
    return fr_from_mont(c, out)


def evaluate_many(curve, jobs, xs) -> list[list[int]]:
    """Several evaluation jobs at up to two points in ONE library call and one wait (mzk_poly_eval_many_dev: what round 4 needs,
    prover.rs:216-299).  jobs: list of (polys_dev, length or None, which[, offset]) with polys_dev, length and offset as in `evaluate`
    and which = index into xs (one or two points).  Returns the values per job as canonical ints."""
    c = _curve(curve)
    assert 1 <= len(xs) <= 2 and len(jobs) <= 64
    xm = np.ascontiguousarray(fr_to_mont(c, [xs[0], xs[-1]]))
    ptrs, lens, batches, strides, which = [], [], [], [], []
    for job in jobs:
        t, length, w = job[:3]
        offset = job[3] if len(job) > 3 else 0
        batch, stride = (1, t.shape[0]) if t.dim() == 2 else (t.shape[0], t.shape[1])
        n = stride - offset if length is None else length
        assert t.is_cuda and t.is_contiguous() and t.shape[-1] == 4 and 0 <= w < len(xs) and 0 <= offset and n >= 0 and offset + n <= stride
        ptrs.append(t.data_ptr() + 32 * offset); lens.append(n); batches.append(batch); strides.append(stride); which.append(w)
    a_ptr = np.array(ptrs, dtype=np.uint64); a_len = np.array(lens, dtype=np.uint64); a_b = np.array(batches, dtype=np.uint32)
    a_s = np.array(strides, dtype=np.uint64); a_w = np.array(which, dtype=np.uint32)
    out = np.empty((int(a_b.sum()), 4), dtype=np.uint64)
    _lib.check(_lib.ensure_init().mzk_poly_eval_many_dev(c.curve_id, len(jobs), C.c_void_p(a_ptr.ctypes.data), C.c_void_p(a_len.ctypes.data),
                                                         C.c_void_p(a_b.ctypes.data), C.c_void_p(a_s.ctypes.data), C.c_void_p(a_w.ctypes.data),
                                                         C.c_void_p(xm.ctypes.data), C.c_void_p(out.ctypes.data), _stream(jobs[0][0], None)),
               "mzk_poly_eval_many_dev")
    vals = fr_from_mont(c, out)
    res, at = [], 0
    for b in batches:
        res.append(vals[at:at + b]); at += b
    return res


def lincomb(curve, terms, out_len: int | None = None, out=None, stream=None):
    """terms: list of (scalar:int, poly:(len,4) CUDA tensor).  Returns sum_k scalar_k * poly_k as a
    (out_len, 4) CUDA tensor (default: the longest input)."""
    import torch
    c = _curve(curve)
    k = len(terms)
    polys = [p for _, p in terms]
    n_out = max(p.shape[0] for p in polys) if out_len is None else out_len
    if out is None:
        out = torch.empty((n_out, 4), dtype=torch.int64, device=polys[0].device)
    ptrs = (C.c_void_p * k)(*[p.data_ptr() for p in polys])
    lens = (C.c_uint64 * k)(*[p.shape[0] for p in polys])
    sc = fr_to_mont(c, [s for s, _ in terms])
    _lib.check(_lib.ensure_init().mzk_poly_lincomb_dev(c.curve_id, k, ptrs, lens, C.c_void_p(sc.ctypes.data), out.data_ptr(), n_out,
                                                       _stream(out, stream)), "mzk_poly_lincomb_dev")
    return out


def div_by_linear(curve, poly_dev, z: int, stream=None, rem_out=None):
    """Quotient of p(X) / (X - z) as a (len-1, 4) CUDA tensor (remainder dropped, as ark-poly does).  rem_out: a (1, 4) CUDA tensor that
    receives the remainder p(z) (asynchronously)."""
    import torch
    c = _curve(curve)
    n = poly_dev.shape[0]
    out = torch.zeros((max(n - 1, 0), 4), dtype=torch.int64, device=poly_dev.device)
    zm = fr_to_mont(c, [z])[0]
    L = _lib.ensure_init()
    if rem_out is None:
        _lib.check(L.mzk_poly_div_linear_dev(c.curve_id, poly_dev.data_ptr(), n, C.c_void_p(zm.ctypes.data), out.data_ptr(), _stream(poly_dev, stream)),
                   "mzk_poly_div_linear_dev")
    else:
        _lib.check(L.mzk_poly_div_linear_rem_dev(c.curve_id, poly_dev.data_ptr(), n, C.c_void_p(zm.ctypes.data), out.data_ptr(), rem_out.data_ptr(),
                                                 _stream(poly_dev, stream)), "mzk_poly_div_linear_rem_dev")
    return out


def div_by_roots_of_unity(curve, poly_dev, log_order: int, first: int, count: int, stream=None):
    """Quotient of p(X) / prod_{i<count} (X - w^(first+i)), w the primitive 2^log_order-th root of unity, as a
    (len - count, 4) CUDA tensor (remainder dropped) -- the division of `compute_linking_quotient`
    (plonk/src/proof_system/proof_linking.rs:119-158).  The call synchronises."""
    import torch
    c = _curve(curve)
    n = poly_dev.shape[0]
    out = torch.zeros((max(n - count, 0), 4), dtype=torch.int64, device=poly_dev.device)
    _lib.check(_lib.ensure_init().mzk_poly_div_roots_dev(c.curve_id, poly_dev.data_ptr(), n, log_order, first, count, out.data_ptr(),
                                                         _stream(poly_dev, stream)), "mzk_poly_div_roots_dev")
    return out


def degree_len_async(poly_dev, stream=None):
    """Number of coefficients up to and including the highest non-zero one (0 for the zero polynomial), as a one-element
    int64 CUDA tensor written asynchronously: read it (`int(t.item())`) after a point where the stream is synchronised
    anyway.  `DensePolynomial::degree` + 1 of the stripped vector (prover.rs:915-918)."""
    import torch
    assert poly_dev.is_cuda and poly_dev.is_contiguous() and poly_dev.shape[-1] == 4
    out = torch.empty((1,), dtype=torch.int64, device=poly_dev.device)
    _lib.check(_lib.ensure_init().mzk_poly_degree_dev(poly_dev.data_ptr(), poly_dev.shape[0], out.data_ptr(), _stream(poly_dev, stream)),
               "mzk_poly_degree_dev")
    return out


def mask(curve, rows, n: int, blinders, stream=None):
    """`Prover::mask_polynomial` (prover.rs:463-486) on device rows, in place: rows[i] (a CUDA tensor view of at least
    n + h slots holding n coefficients) += (b_0 + b_1 X + .. )(X^n - 1) with blinders[i] = [b_0, .., b_{h-1}] (Python ints)."""
    c = _curve(curve)
    k, h = len(rows), len(blinders[0])
    assert k == len(blinders) and all(len(b) == h for b in blinders) and all(r.is_cuda and r.shape[0] >= n + h for r in rows)
    bm = fr_to_mont(c, [x for b in blinders for x in b])
    ptrs = (C.c_void_p * k)(*[r.data_ptr() for r in rows])
    _lib.check(_lib.ensure_init().mzk_poly_mask_dev(c.curve_id, k, ptrs, n, h, C.c_void_p(bm.ctypes.data), _stream(rows[0], stream)), "mzk_poly_mask_dev")


def gather_witness(witness_dev, wire_variables_dev, out=None, stream=None):
    """out[i, j] = witness[wire_variables[i, j]]: the gather of Arithmetization::compute_wire_polynomials
    (relation/src/constraint_system.rs:1225-1247).  witness (n_vars, 4) int64, wire_variables (W, n) int32, both CUDA."""
    import torch
    W, n = wire_variables_dev.shape
    assert wire_variables_dev.dtype == torch.int32 and wire_variables_dev.is_contiguous() and witness_dev.is_contiguous()
    if out is None:
        out = torch.empty((W, n, 4), dtype=torch.int64, device=witness_dev.device)
    _lib.check(_lib.ensure_init().mzk_plonk_gather_witness_dev(witness_dev.data_ptr(), witness_dev.shape[0], wire_variables_dev.data_ptr(), W * n,
                                                               out.data_ptr(), _stream(out, stream)), "mzk_plonk_gather_witness_dev")
    return out
