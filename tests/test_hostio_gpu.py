"""GPU: the host-pointer entry points (mzk_ntt, mzk_ntt_batch, mzk_msm, mzk_msm_batch) on their I/O slots -- page-locked and
pageable host memory, batches longer than the pipeline, and concurrent callers (the reference transforms and commits from Rayon
workers: prover.rs:552-562, univariate_kzg/mod.rs:125-127) -- against the C restatement."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pinned(L, rows):
    p = C.c_void_p()
    assert L.mzk_host_alloc(rows * 32, C.byref(p)) == 0
    a = np.ctypeslib.as_array((C.c_uint64 * (rows * 4)).from_address(p.value)).reshape(rows, 4)
    return p, a


def test_ntt_batch_longer_than_the_pipeline_pinned_and_pageable(gpu, mj, cref):
    L = mj.load()
    c = mj.params.BLS12_381
    log_n, N = 13, 1 << 13
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    lens = [N, N - 1, 5, 1, N // 8 + 3, N, 77, N // 2, 3]                      # 9 polynomials > the 3 pipeline slots
    src = [mj.params.random_fr_mont(c, ln, seed=100 + i) for i, ln in enumerate(lens)]
    want = [cref.ntt(0, np.concatenate([s, np.zeros((N - len(s), 4), dtype=np.uint64)]), log_n, False, g, threads=2) for s in src]
    for pinned in (False, True):
        handles, bufs = [], []
        for s in src:
            if pinned:
                p, a = _pinned(L, N)
                handles.append(p)
            else:
                a = np.zeros((N, 4), dtype=np.uint64)
            a[:len(s)] = s
            a[len(s):] = 0xdeadbeef                                              # beyond in_len: must be ignored, not transformed
            bufs.append(a)
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        ln = (C.c_uint64 * len(bufs))(*lens)
        assert L.mzk_ntt_batch(0, len(bufs), ptrs, ln, log_n, 0, g.ctypes.data_as(C.c_void_p)) == 0, L.mzk_last_error()
        for b, w in zip(bufs, want):
            assert np.array_equal(b, w)
        # and back: inverse coset transform of the whole batch
        full = (C.c_uint64 * len(bufs))(*[N] * len(bufs))
        assert L.mzk_ntt_batch(0, len(bufs), ptrs, full, log_n, 1, g.ctypes.data_as(C.c_void_p)) == 0
        for b, s in zip(bufs, src):
            assert np.array_equal(b[:len(s)], s) and not b[len(s):].any()
        for p in handles:
            assert L.mzk_host_free(p) == 0


@pytest.mark.parametrize("curve_id", [0, 1])
def test_ntt_batch_on_the_collaborative_provers_public_polynomials(gpu, mj, cref, curve_id):
    """the multiprover transforms its PUBLIC polynomials -- 13 selectors and 5 sigmas of n coefficients each -- onto
    the GENERATOR coset of the 8n-point quotient domain, one `coset.fft(poly.coeffs())` per polynomial
    (plonk/src/multiprover/proof_system/prover.rs:364-371).  The swap is ONE mzk_ntt_batch call with in_lens = n: every row against the C
    restatement of ark-poly's transform of the zero-padded coefficients (what `fft` does with a short slice)."""
    L = mj.load()
    c = mj.params.BLS12_381 if curve_id == 0 else mj.params.BN254
    n, log_m = 1 << 9, 12                                                        # quot_domain = 8n
    M = 1 << log_m
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    polys = [mj.params.random_fr_mont(c, n, seed=700 + 31 * curve_id + i) for i in range(13 + 5)]
    polys[3][n - 5:] = 0                                                         # a selector of lower degree: trailing zero coefficients
    polys[7][:] = 0                                                              # an unused selector: the zero polynomial
    bufs = []
    for s in polys:
        a = np.empty((M, 4), dtype=np.uint64)
        a[:n] = s
        a[n:] = 0x5a5a5a5a5a5a5a5a                                              # beyond in_len: ignored
        bufs.append(a)
    ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    lens = (C.c_uint64 * len(bufs))(*[n] * len(bufs))
    assert L.mzk_ntt_batch(curve_id, len(bufs), ptrs, lens, log_m, 0, g.ctypes.data_as(C.c_void_p)) == 0, L.mzk_last_error()
    for s, b in zip(polys, bufs):
        padded = np.concatenate([s, np.zeros((M - n, 4), dtype=np.uint64)])
        assert np.array_equal(b, cref.ntt(curve_id, padded, log_m, False, g, threads=2))
    assert not bufs[7].any()


def test_host_register_and_error_paths(gpu, mj):
    L = mj.load()
    a = np.zeros((1 << 12, 4), dtype=np.uint64)
    assert L.mzk_host_register(C.c_void_p(a.ctypes.data), a.nbytes) == 0
    a[:] = mj.params.random_fr_mont(mj.params.BN254, 1 << 12, seed=3)
    keep = a.copy()
    assert L.mzk_ntt(1, C.c_void_p(a.ctypes.data), 1 << 12, 12, 0, None) == 0
    assert L.mzk_ntt(1, C.c_void_p(a.ctypes.data), 1 << 12, 12, 1, None) == 0
    assert np.array_equal(a, keep)
    assert L.mzk_host_unregister(C.c_void_p(a.ctypes.data)) == 0
    assert L.mzk_host_alloc(16, None) == -1 and L.mzk_host_register(None, 16) == -1
    ptrs = (C.c_void_p * 2)(a.ctypes.data, None)
    ln = (C.c_uint64 * 2)(4, 4)
    assert L.mzk_ntt_batch(1, 2, ptrs, ln, 2, 0, None) == -1 and np.array_equal(a, keep), "a null entry fails the call before anything is transformed"


@pytest.mark.parametrize("curve_id", [0, 1])
def test_concurrent_host_pointer_callers(gpu, mj, cref, curve_id):
    """Eight threads issue mzk_ntt / mzk_msm / mzk_msm_batch with host pointers at once (more threads than I/O slots)."""
    L = mj.load()
    c = mj.params.CURVES[curve_id]
    n = 3000
    bases = cref.g1_arith_bases(curve_id, 0xabc + curve_id, 0x77, n)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    log_n, N = 11, 1 << 11
    errors = []

    def worker(t):
        try:
            for it in range(4):
                seed = 1000 * t + it
                if (t + it) % 2 == 0:
                    x = mj.params.random_fr_mont(c, N, seed=seed)
                    want = cref.ntt(curve_id, x, log_n, False, None, threads=1)
                    buf = x.copy()
                    rc = L.mzk_ntt(curve_id, C.c_void_p(buf.ctypes.data), N, log_n, 0, None)
                    if rc != 0 or not np.array_equal(buf, want):
                        errors.append(("ntt", t, it, rc))
                else:
                    ln = 1 + (seed * 37) % n
                    sc = mj.params.random_fr_mont(c, ln, seed=seed)
                    want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[:ln], sc, scalars_are_mont=True, threads=1))[0]
                    fl = c.fq_limbs
                    if it % 2:
                        out = np.zeros(3 * fl, dtype=np.uint64)
                        rc = L.mzk_msm(pp.handle, 0, C.c_void_p(sc.ctypes.data), ln, 1, out.ctypes.data_as(C.c_void_p))
                    else:
                        outs = np.zeros((2, 3 * fl), dtype=np.uint64)
                        ptrs = (C.c_void_p * 2)(sc.ctypes.data, sc.ctypes.data)
                        lens = (C.c_uint64 * 2)(ln, ln)
                        rc = L.mzk_msm_batch(pp.handle, 2, ptrs, lens, None, 1, outs.ctypes.data_as(C.c_void_p))
                        out = outs[1]                                             # (the Jacobian representatives of equal points may differ)
                        if not np.array_equal(cref.jac_to_affine(curve_id, outs[0].reshape(1, 3, fl))[0], want):
                            errors.append(("msm_batch[0]", t, it, rc))
                    got = cref.jac_to_affine(curve_id, out.reshape(1, 3, fl))[0]
                    if rc != 0 or not np.array_equal(got, want):
                        errors.append(("msm", t, it, rc))
        except Exception as e:                                                     # noqa: BLE001
            errors.append(("exc", t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    pp.release()
