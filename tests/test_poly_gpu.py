"""GPU parity for the dense-polynomial primitives of prover rounds 4-5 (evaluate, linear combination,
division by X - z) against the C oracle and big-int checks."""
import random

import numpy as np
import pytest

from conftest import fr_from_mont_limbs, fr_mont_limbs

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 7, 255, 256, 257, 16384, 16385, 100003, (1 << 20) + 3])
def test_evaluate_matches_oracle(gpu, mj, cref, curve_id, n):
    c = mj.params.CURVES[curve_id]
    rng = random.Random(n)
    batch = 3 if n < 200000 else 2
    polys = mj.params.random_fr_mont(c, batch * n, seed=n % 1000).reshape(batch, n, 4)
    for x in (rng.randrange(c.r), 0, 1, c.r - 1):
        got = mj.poly.evaluate(c, _dev(polys), x)
        xm = mj.params.fr_to_mont(c, [x])[0]
        want = [mj.params.fr_from_mont(c, cref.poly_eval(curve_id, polys[b], xm).reshape(1, 4))[0] for b in range(batch)]
        assert got == want, (n, x)
        if n > 50000:
            break
    # a shorter logical length inside a wider slab (zero padding is NOT assumed)
    if n >= 7:
        got = mj.poly.evaluate(c, _dev(polys), 5, length=n - 3)
        xm = mj.params.fr_to_mont(c, [5])[0]
        assert got[0] == mj.params.fr_from_mont(c, cref.poly_eval(curve_id, polys[0, :n - 3], xm).reshape(1, 4))[0]


@pytest.mark.parametrize("curve_id", [0, 1])
def test_evaluate_many_matches_the_oracle_and_single_calls(gpu, mj, cref, curve_id):
    """mzk_poly_eval_many_dev (a round's evaluations at zeta and zeta * omega in one call) against Horner's rule on the C oracle -- every
    value of every job -- and against mzk_poly_eval_dev job by job: batches, shorter logical lengths, an empty job, one point only."""
    import ctypes as C
    from mpc_jellyfish_amd import lib as mlib
    c = mj.params.CURVES[curve_id]
    rng = random.Random(5)
    a = _dev(mj.params.random_fr_mont(c, 3 * 5000, seed=1).reshape(3, 5000, 4))
    b = _dev(mj.params.random_fr_mont(c, 70001, seed=2))
    d = _dev(mj.params.random_fr_mont(c, 2 * 300, seed=3).reshape(2, 300, 4))
    x0, x1 = rng.randrange(c.r), rng.randrange(c.r)
    jobs = [(a, None, 0), (b, None, 1), (d, 257, 1), (b, 0, 0), (a, 4097, 1), (d, None, 0)]
    got = mj.poly.evaluate_many(c, jobs, [x0, x1])
    xm = [mj.params.fr_to_mont(c, [x])[0] for x in (x0, x1)]
    for (t, length, w), g in zip(jobs, got):
        assert g == mj.poly.evaluate(c, t, [x0, x1][w], length=length)
        host = t.cpu().numpy().view(np.uint64)
        host = host.reshape(1, -1, 4) if host.ndim == 2 else host
        ln = host.shape[1] if length is None else length
        want = [mj.params.fr_from_mont(c, cref.poly_eval(curve_id, row[:ln], xm[w]).reshape(1, 4))[0] if ln else 0 for row in host]
        assert g == want, (ln, w)
    assert mj.poly.evaluate_many(c, [(b, None, 0)], [x1])[0] == mj.poly.evaluate(c, b, x1)
    L = mlib.ensure_init()
    assert L.mzk_poly_eval_many_dev(c.curve_id, 65, None, None, None, None, None, None, None, None) == -1
    assert L.mzk_poly_eval_many_dev(c.curve_id, 0, None, None, None, None, None, None, None, None) == 0


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 3, 2047, 2048, 2049, 70001, (1 << 20) + 3])
def test_div_by_linear_matches_oracle(gpu, mj, cref, curve_id, n):
    c = mj.params.CURVES[curve_id]
    rng = random.Random(n + 1)
    p = mj.params.random_fr_mont(c, n, seed=n % 997)
    for z in (rng.randrange(1, c.r), 1, c.r - 1, 0):
        got = mj.poly.div_by_linear(c, _dev(p), z).cpu().numpy().view(np.uint64)
        want = cref.poly_div_linear(curve_id, p, mj.params.fr_to_mont(c, [z])[0])
        assert np.array_equal(got, want), (n, z)
    if 3 <= n <= 3000:
        # exact division: p = (X - z) * q  =>  quotient recovers q
        z = rng.randrange(1, c.r)
        q = [rng.randrange(c.r) for _ in range(n - 1)]
        prod = [(-z * q[0]) % c.r] + [(q[i - 1] - z * q[i]) % c.r for i in range(1, n - 1)] + [q[n - 2]]
        got = mj.poly.div_by_linear(c, _dev(fr_mont_limbs(c, prod)), z).cpu().numpy().view(np.uint64)
        assert fr_from_mont_limbs(c, got) == q


@pytest.mark.parametrize("curve_id", [0, 1])
def test_lincomb_matches_oracle(gpu, mj, cref, curve_id):
    c = mj.params.CURVES[curve_id]
    rng = random.Random(77)
    lens = [1000, 1003, 1, 999, 1003, 500, 1002]
    polys = [mj.params.random_fr_mont(c, n, seed=n + i) for i, n in enumerate(lens)]
    scalars = [rng.randrange(c.r) for _ in lens]
    scalars[2] = 0
    scalars[3] = 1
    for out_len in (1003, 600, 1200):
        got = mj.poly.lincomb(c, list(zip(scalars, [_dev(p) for p in polys])), out_len=out_len).cpu().numpy().view(np.uint64)
        want = cref.poly_lincomb(curve_id, polys, mj.params.fr_to_mont(c, scalars), out_len)
        assert np.array_equal(got, want), out_len
    # in place: out aliases the first input (r_quot = r_quot + coeff * poly, prover.rs:350-353)
    d = [_dev(p) for p in polys]
    acc = d[1].clone()
    mj.poly.lincomb(c, [(1, acc), (scalars[4], d[4])], out=acc)
    want = cref.poly_lincomb(curve_id, [polys[1], polys[4]], mj.params.fr_to_mont(c, [1, scalars[4]]), 1003)
    assert np.array_equal(acc.cpu().numpy().view(np.uint64), want)
    with pytest.raises(mj.MzkError):
        mj.poly.lincomb(c, [(1, d[0])] * 33)


def test_degree_len(gpu, mj):
    """mzk_poly_degree_dev: number of coefficients up to the highest non-zero one (`DensePolynomial::degree` + 1 after trailing zeros
    are stripped), the guard behind WrongQuotientPolyDegree (prover.rs:915-918)."""
    import torch
    c = mj.params.BLS12_381
    for n, top in ((1, 0), (5, 4), (4097, 0), (4097, 4096), (100000, 77777), (1 << 20, (1 << 20) - 3)):
        a = np.zeros((n, 4), dtype=np.uint64)
        a[:top + 1] = mj.params.random_fr_mont(c, top + 1, seed=n)
        a[top] = mj.params.fr_to_mont(c, [3])[0]                       # make sure the top coefficient is non-zero
        t = torch.from_numpy(a.view(np.int64)).cuda()
        assert int(mj.poly.degree_len_async(t).item()) == top + 1
        assert int(mj.poly.degree_len_async(t[top + 1:]).item()) == 0 if top + 1 < n else True
    z = torch.zeros((1000, 4), dtype=torch.int64, device="cuda")
    assert int(mj.poly.degree_len_async(z).item()) == 0
    z[999, 3] = 1                                                       # any non-zero limb counts
    assert int(mj.poly.degree_len_async(z).item()) == 1000
    assert int(mj.poly.degree_len_async(z[:0]).item()) == 0


def test_gather_witness_and_stream_helpers(gpu, mj):
    """mzk_plonk_gather_witness_dev (the gather of compute_wire_polynomials, relation/src/constraint_system.rs:1225-1247): ragged counts,
    repeated and out-of-range variable indices (an index >= n_vars yields zero); and the stream helpers a host-resident witness is
    uploaded with (mzk_stream_create / wait_stream / dev_upload_async / stream_sync)."""
    import ctypes as C
    import torch
    c = mj.params.BLS12_381
    L = mj.load()
    for n_vars, count in ((1, 1), (7, 130), (1000, 4099)):
        wit = mj.params.random_fr_mont(c, n_vars, seed=n_vars)
        rs = np.random.default_rng(count)
        idx = rs.integers(0, n_vars, size=count).astype(np.uint32)
        idx[::17] = n_vars + rs.integers(0, 5, size=len(idx[::17])).astype(np.uint32)          # out of range -> zero
        d_idx = torch.from_numpy(idx.view(np.int32)).cuda()
        host = torch.from_numpy(wit.view(np.int64)).pin_memory()
        d_wit = torch.empty_like(host, device="cuda")
        st = C.c_void_p()
        mj.lib.check(L.mzk_stream_create(C.byref(st)), "mzk_stream_create")
        main = torch.cuda.current_stream().cuda_stream
        mj.lib.check(L.mzk_dev_upload_async(d_wit.data_ptr(), host.data_ptr(), host.numel() * 8, st), "upload")
        mj.lib.check(L.mzk_stream_wait_stream(main, st), "wait")
        out = torch.empty((count, 4), dtype=torch.int64, device="cuda")
        mj.lib.check(L.mzk_plonk_gather_witness_dev(d_wit.data_ptr(), n_vars, d_idx.data_ptr(), count, out.data_ptr(), main), "gather")
        torch.cuda.synchronize()
        mj.lib.check(L.mzk_stream_sync(st), "sync")
        mj.lib.check(L.mzk_stream_destroy(st), "destroy")
        want = np.zeros((count, 4), dtype=np.uint64)
        ok = idx < n_vars
        want[ok] = wit[idx[ok]]
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    dev = C.c_int32(-7)
    assert L.mzk_get_device(C.byref(dev)) == 0 and dev.value == 0


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("n,W", [(4, 5), (8, 5), (16, 6), (1000, 5), (1 << 12, 6), ((1 << 16), 5), (37, 2), (64, 8)])
def test_split_quotient_matches_the_reference_rule(gpu, mj, curve_id, n, W):
    """mzk_poly_split_quotient_dev (round 3's split in one launch) against `split_quotient_polynomial` restated with Python integers
    (plonk/src/proof_system/prover.rs:902-960): slices of n + 2 coefficients, the last one what is left up to degree W (n + 1) + 2,
    slice i < W - 1 gains b_i X^(n+2), slice i > 0 loses b_{i-1} from its constant term; everything else in a row of `stride` slots is zero --
    whatever the rows held before."""
    import ctypes as C
    import torch
    from mpc_jellyfish_amd import lib as mlib
    c = mj.params.CURVES[curve_id]
    rng = random.Random(n * 31 + W)
    expected = W * (n + 1) + 2
    qi = [rng.randrange(c.r) for _ in range(expected + 1)] + [rng.randrange(c.r) for _ in range(5)]     # (what lies beyond the degree is not read)
    b = [rng.randrange(c.r) for _ in range(W - 1)]
    L = mlib.ensure_init()
    for stride in (n + 3, n + 7):
        d_q = _dev(fr_mont_limbs(c, qi))
        d_out = _dev(mj.params.random_fr_mont(c, W * stride, seed=9))                                  # dirty rows
        bl = fr_mont_limbs(c, b)
        mlib.check(L.mzk_poly_split_quotient_dev(curve_id, C.c_void_p(d_q.data_ptr()), n, W, C.c_void_p(bl.ctypes.data), C.c_void_p(d_out.data_ptr()), stride, None),
                   "mzk_poly_split_quotient_dev")
        torch.cuda.synchronize()
        got = fr_from_mont_limbs(c, d_out.cpu().numpy().view(np.uint64).reshape(-1, 4))
        for i in range(W):
            lo, hi = i * (n + 2), ((i + 1) * (n + 2) if i < W - 1 else expected + 1)
            want = qi[lo:hi] + [0] * (stride - (hi - lo))
            if i < W - 1:
                want[n + 2] = b[i]
            if i > 0:
                want[0] = (want[0] - b[i - 1]) % c.r
            assert got[i * stride:(i + 1) * stride] == want, (n, W, stride, i)
    # argument checks: too many parts, rows too short
    assert L.mzk_poly_split_quotient_dev(curve_id, C.c_void_p(d_q.data_ptr()), n, 9, C.c_void_p(bl.ctypes.data), C.c_void_p(d_out.data_ptr()), n + 3, None) == -1
    assert L.mzk_poly_split_quotient_dev(curve_id, C.c_void_p(d_q.data_ptr()), n, W, C.c_void_p(bl.ctypes.data), C.c_void_p(d_out.data_ptr()), n + 2, None) == -1


def test_memset2d(gpu, mj):
    """mzk_dev_memset2d: `height` runs of `width` bytes, `pitch` apart, nothing else touched."""
    import ctypes as C
    import torch
    from mpc_jellyfish_amd import lib as mlib
    L = mlib.ensure_init()
    a = torch.full((7, 100), 0x55, dtype=torch.uint8, device="cuda")
    mlib.check(L.mzk_dev_memset2d(C.c_void_p(a.data_ptr() + 90), 100, 0, 10, 7, None), "mzk_dev_memset2d")
    torch.cuda.synchronize()
    h = a.cpu().numpy()
    assert (h[:, :90] == 0x55).all() and (h[:, 90:] == 0).all()
    assert L.mzk_dev_memset2d(C.c_void_p(a.data_ptr()), 5, 0, 10, 7, None) == -1
    assert L.mzk_dev_memset2d(None, 100, 0, 0, 7, None) == 0
