"""oracle/cref.py -- TEST INFRASTRUCTURE ONLY: ctypes wrapper around oracle/_build/liboracle.so
(the C restatement in cpu_ref.c).  numpy uint64 arrays in, numpy uint64 arrays out.
May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "_build", "liboracle.so")
    if force or not os.path.exists(path):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return path


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u64p = C.POINTER(C.c_uint64)
        L.orc_ntt.argtypes = [C.c_int, u64p, C.c_int, C.c_int, u64p, C.c_int]
        L.orc_fr_convert.argtypes = [C.c_int, u64p, u64p, C.c_size_t, C.c_int]
        L.orc_fq_convert.argtypes = [C.c_int, u64p, u64p, C.c_size_t, C.c_int]
        L.orc_fr_mul.argtypes = [C.c_int, u64p, u64p, u64p, C.c_size_t]
        L.orc_poly_eval.argtypes = [C.c_int, u64p, C.c_size_t, u64p, u64p]
        L.orc_domain_element.argtypes = [C.c_int, C.c_int, C.c_uint64, u64p, u64p]
        L.orc_msm.argtypes = [C.c_int, u64p, u64p, C.c_size_t, C.c_int, u64p, C.c_int, C.c_int]
        L.orc_jac_to_affine.argtypes = [C.c_int, u64p, u64p, C.c_size_t]
        L.orc_g1_mul_gen.argtypes = [C.c_int, u64p, u64p]
        L.orc_g1_mul.argtypes = [C.c_int, u64p, u64p, u64p]
        L.orc_g1_arith_bases.argtypes = [C.c_int, u64p, u64p, C.c_size_t, u64p]
        L.orc_srs_powers.argtypes = [C.c_int, u64p, C.c_size_t, u64p, C.c_int]
        L.orc_g1_count_off_curve.argtypes = [C.c_int, u64p, C.c_size_t]
        L.orc_g1_count_off_curve.restype = C.c_long
        L.orc_plookup_sorted.restype = C.c_long
        L.orc_plookup_merge.argtypes = [C.c_int, C.c_size_t, u64p, u64p, u64p, u64p, u64p, u64p]
        L.orc_plookup_sorted.argtypes = [C.c_int, C.c_size_t, u64p, u64p, u64p]
        L.orc_plookup_product.argtypes = [C.c_int, C.c_int, u64p, u64p, u64p, u64p, u64p, u64p, C.c_int]
        L.orc_plonk_quotient_ultra.argtypes = [C.c_int, C.c_int, u64p, C.c_size_t, u64p, u64p, u64p, u64p, u64p, u64p, C.c_int]
        L.orc_plonk_quotient.argtypes = [C.c_int, C.c_int, C.c_int, u64p, C.c_size_t, u64p, u64p, u64p, u64p, u64p, C.c_int]
        L.orc_plonk_perm_product.argtypes = [C.c_int, C.c_int, C.c_int, u64p, u64p, u64p, u64p, u64p, u64p, C.c_int]
        L.orc_poly_div_linear.argtypes = [C.c_int, u64p, C.c_size_t, u64p, u64p]
        L.orc_poly_lincomb.argtypes = [C.c_int, C.c_int, u64p, C.c_size_t, C.POINTER(C.c_size_t), u64p, u64p, C.c_size_t]
        _LIB = L
    return _LIB


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _chk(rc):
    if rc != 0:
        raise RuntimeError(f"oracle call failed rc={rc}")


def fq_limbs(curve: int) -> int:
    return 6 if curve == 0 else 4


def ints_to_limbs(vals, n_limbs: int) -> np.ndarray:
    out = np.zeros((len(vals), n_limbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(n_limbs):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a: np.ndarray) -> list[int]:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, a.shape[-1])
    return [sum(int(a[i, j]) << (64 * j) for j in range(a.shape[1])) for i in range(a.shape[0])]


def ntt(curve: int, data_mont: np.ndarray, log_n: int, inverse: bool = False, coset_mont=None, threads: int = 1):
    """data_mont: (N,4) uint64 Montgomery, N = 2^log_n; returns a new array."""
    a = np.ascontiguousarray(data_mont, dtype=np.uint64).copy()
    assert a.shape == (1 << log_n, 4)
    cp = _p(np.ascontiguousarray(coset_mont, dtype=np.uint64)) if coset_mont is not None else None
    _chk(lib().orc_ntt(curve, _p(a), log_n, int(inverse), cp, threads))
    return a


def fr_convert(curve: int, a: np.ndarray, to_mont: bool) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    _chk(lib().orc_fr_convert(curve, _p(a), _p(out), a.size // 4, int(to_mont)))
    return out


def fq_convert(curve: int, a: np.ndarray, to_mont: bool) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    _chk(lib().orc_fq_convert(curve, _p(a), _p(out), a.size // fq_limbs(curve), int(to_mont)))
    return out


def fr_mul(curve: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    _chk(lib().orc_fr_mul(curve, _p(a), _p(b), _p(out), a.size // 4))
    return out


def poly_eval(curve: int, coeffs_mont: np.ndarray, x_mont: np.ndarray) -> np.ndarray:
    c = np.ascontiguousarray(coeffs_mont, dtype=np.uint64)
    x = np.ascontiguousarray(x_mont, dtype=np.uint64)
    out = np.empty(4, dtype=np.uint64)
    _chk(lib().orc_poly_eval(curve, _p(c), c.size // 4, _p(x), _p(out)))
    return out


def domain_element(curve: int, log_n: int, k: int, coset_mont=None) -> np.ndarray:
    out = np.empty(4, dtype=np.uint64)
    cp = _p(np.ascontiguousarray(coset_mont, dtype=np.uint64)) if coset_mont is not None else None
    _chk(lib().orc_domain_element(curve, log_n, k, cp, _p(out)))
    return out


def msm(curve: int, bases_xy: np.ndarray, scalars: np.ndarray, scalars_are_mont: bool = False,
        threads: int = 1, window_bits: int = 0) -> np.ndarray:
    """Returns Jacobian (3, fq_limbs) Montgomery."""
    L = fq_limbs(curve)
    b = np.ascontiguousarray(bases_xy, dtype=np.uint64)
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = min(b.size // (2 * L), s.size // 4)
    out = np.empty((3, L), dtype=np.uint64)
    _chk(lib().orc_msm(curve, _p(b), _p(s), n, int(scalars_are_mont), _p(out), threads, window_bits))
    return out


def jac_to_affine(curve: int, xyz: np.ndarray) -> np.ndarray:
    L = fq_limbs(curve)
    a = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(-1, 3, L)
    out = np.empty((a.shape[0], 2, L), dtype=np.uint64)
    _chk(lib().orc_jac_to_affine(curve, _p(a), _p(out), a.shape[0]))
    return out


def g1_mul_gen(curve: int, k: int) -> np.ndarray:
    out = np.empty((2, fq_limbs(curve)), dtype=np.uint64)
    _chk(lib().orc_g1_mul_gen(curve, _p(ints_to_limbs([k], 4)), _p(out)))
    return out


def g1_mul(curve: int, p_xy: np.ndarray, k: int) -> np.ndarray:
    out = np.empty((2, fq_limbs(curve)), dtype=np.uint64)
    _chk(lib().orc_g1_mul(curve, _p(np.ascontiguousarray(p_xy, dtype=np.uint64)), _p(ints_to_limbs([k], 4)), _p(out)))
    return out


def g1_arith_bases(curve: int, s: int, t: int, n: int) -> np.ndarray:
    out = np.empty((n, 2, fq_limbs(curve)), dtype=np.uint64)
    _chk(lib().orc_g1_arith_bases(curve, _p(ints_to_limbs([s], 4)), _p(ints_to_limbs([t], 4)), n, _p(out)))
    return out


def srs_powers(curve: int, beta: int, n: int, threads: int = 1) -> np.ndarray:
    out = np.empty((n, 2, fq_limbs(curve)), dtype=np.uint64)
    _chk(lib().orc_srs_powers(curve, _p(ints_to_limbs([beta], 4)), n, _p(out), threads))
    return out


def count_off_curve(curve: int, xy: np.ndarray) -> int:
    a = np.ascontiguousarray(xy, dtype=np.uint64)
    return int(lib().orc_g1_count_off_curve(curve, _p(a), a.size // (2 * fq_limbs(curve))))


def plonk_quotient(curve: int, log_n: int, polys: np.ndarray, k_mont: np.ndarray, alpha, beta, gamma, threads: int = 1) -> np.ndarray:
    """polys: (13 + 2*5 + 2, poly_len, 4) Montgomery coefficients (selectors, sigmas, wires, z, pi);
    returns the 8n quotient coefficients (prover.rs:512-673, one TurboPlonk instance)."""
    p = np.ascontiguousarray(polys, dtype=np.uint64)
    assert p.ndim == 3 and p.shape[0] == 25 and p.shape[2] == 4
    out = np.empty((8 << log_n, 4), dtype=np.uint64)
    args = [np.ascontiguousarray(a, dtype=np.uint64) for a in (k_mont, alpha, beta, gamma)]
    _chk(lib().orc_plonk_quotient(curve, log_n, 5, _p(p), p.shape[1], _p(args[0]), _p(args[1]), _p(args[2]), _p(args[3]), _p(out), threads))
    return out


def plonk_perm_product(curve: int, log_n: int, wires: np.ndarray, sigma_vals: np.ndarray, k_mont: np.ndarray, beta, gamma, threads: int = 1) -> np.ndarray:
    """wires, sigma_vals: (5, n, 4) Montgomery values; returns the n coefficients of the permutation
    product polynomial (constraint_system.rs:1197-1223)."""
    w = np.ascontiguousarray(wires, dtype=np.uint64)
    sg = np.ascontiguousarray(sigma_vals, dtype=np.uint64)
    out = np.empty((1 << log_n, 4), dtype=np.uint64)
    args = [np.ascontiguousarray(a, dtype=np.uint64) for a in (k_mont, beta, gamma)]
    _chk(lib().orc_plonk_perm_product(curve, log_n, w.shape[0], _p(w), _p(sg), _p(args[0]), _p(args[1]), _p(args[2]), _p(out), threads))
    return out


def plookup_merge(curve: int, wires: np.ndarray, tabs: np.ndarray, q_lookup: np.ndarray, tau) -> tuple[np.ndarray, np.ndarray]:
    """constraint_system.rs:1290-1309, 1441-1480: wires (6, n, 4), tabs (4, n, 4) = range, key, table_dom_sep, q_dom_sep,
    q_lookup (n, 4) values -> (merged table, merged lookup witness), each (n, 4)."""
    w, t, q = (np.ascontiguousarray(a, dtype=np.uint64) for a in (wires, tabs, q_lookup))
    n = q.shape[0]
    table, lookup = np.empty((n, 4), dtype=np.uint64), np.empty((n, 4), dtype=np.uint64)
    _chk(lib().orc_plookup_merge(curve, n, _p(w), _p(t), _p(q), _p(np.ascontiguousarray(tau, dtype=np.uint64)), _p(table), _p(lookup)))
    return table, lookup


def plookup_sorted(curve: int, table: np.ndarray, lookup: np.ndarray):
    """constraint_system.rs:1370-1408: the merged sorted vector, or None when a lookup value is not in the table."""
    t, l = np.ascontiguousarray(table, dtype=np.uint64), np.ascontiguousarray(lookup, dtype=np.uint64)
    n = t.shape[0]
    out = np.empty((2 * n, 4), dtype=np.uint64)
    ln = lib().orc_plookup_sorted(curve, n, _p(t), _p(l), _p(out))
    return None if ln < 0 else out[:ln]


def plookup_product(curve: int, log_n: int, table, lookup, sorted_vec, beta, gamma, threads: int = 1) -> np.ndarray:
    """constraint_system.rs:1311-1368: the n coefficients of the Plookup product polynomial."""
    args = [np.ascontiguousarray(a, dtype=np.uint64) for a in (table, lookup, sorted_vec, beta, gamma)]
    out = np.empty((1 << log_n, 4), dtype=np.uint64)
    _chk(lib().orc_plookup_product(curve, log_n, _p(args[0]), _p(args[1]), _p(args[2]), _p(args[3]), _p(args[4]), _p(out), threads))
    return out


def plonk_quotient_ultra(curve: int, log_n: int, polys: np.ndarray, k_mont, tau, alpha, beta, gamma, threads: int = 1) -> np.ndarray:
    """polys: (35, poly_len, 4): selectors[14], sigmas[6], tables[4], wires[6], z, pi, h_1, h_2, prod_lookup (prover.rs:512-888)."""
    p = np.ascontiguousarray(polys, dtype=np.uint64)
    assert p.ndim == 3 and p.shape[0] == 35 and p.shape[2] == 4
    out = np.empty((8 << log_n, 4), dtype=np.uint64)
    args = [np.ascontiguousarray(a, dtype=np.uint64) for a in (k_mont, tau, alpha, beta, gamma)]
    _chk(lib().orc_plonk_quotient_ultra(curve, log_n, _p(p), p.shape[1], *[_p(a) for a in args], _p(out), threads))
    return out


def poly_div_linear(curve: int, coeffs: np.ndarray, z_mont: np.ndarray) -> np.ndarray:
    """Quotient of p(X) / (X - z): (len-1, 4)."""
    c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((max(c.shape[0] - 1, 0), 4), dtype=np.uint64)
    if c.shape[0] >= 2:
        _chk(lib().orc_poly_div_linear(curve, _p(c), c.shape[0], _p(np.ascontiguousarray(z_mont, dtype=np.uint64)), _p(out)))
    return out


def poly_lincomb(curve: int, polys, scalars_mont: np.ndarray, out_len: int) -> np.ndarray:
    """sum_k scalars[k] * polys[k], polys a list of (len_k, 4) arrays."""
    stride = max([p.shape[0] for p in polys] + [1])
    slab = np.zeros((len(polys), stride, 4), dtype=np.uint64)
    lens = (C.c_size_t * len(polys))()
    for k, p in enumerate(polys):
        slab[k, :p.shape[0]] = p
        lens[k] = p.shape[0]
    out = np.empty((out_len, 4), dtype=np.uint64)
    _chk(lib().orc_poly_lincomb(curve, len(polys), _p(slab), stride, lens, _p(np.ascontiguousarray(scalars_mont, dtype=np.uint64)), _p(out), out_len))
    return out
