"""Entry points of include/mzk.h that no host of this repo calls (found by listing the header's symbols against their callers): a caller
of the C ABI may -- the reference-side binding of INTEGRATION.md registers its CommitKey from wherever it lives and may combine all eight
residue classes of the quotient domain itself.  Each is checked against the oracle / the big-int definition."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve_id", [0, 1])
def test_srs_register_dev_matches_host_registration_and_the_oracle(gpu, mj, cref, curve_id):
    """mzk_srs_register_dev: the CommitKey's points (univariate_kzg/srs.rs:36-40) already on the device.  Same MSM results as the
    host-pointer registration and as the C oracle's Pippenger; an empty key registers and commits to infinity."""
    import torch
    c = mj.params.CURVES[curve_id]
    L = mj.load()
    n = 5000
    bases = cref.g1_arith_bases(curve_id, 0x5151 + curve_id, 0x33, n)
    scalars = mj.params.random_fr_mont(c, n, seed=77)
    d_bases = torch.from_numpy(np.ascontiguousarray(bases).view(np.int64)).cuda()
    side = torch.cuda.Stream()
    h = C.c_uint64()
    with torch.cuda.stream(side):
        gpu.check(L.mzk_srs_register_dev(curve_id, C.c_void_p(d_bases.data_ptr()), n, C.byref(h), C.c_void_p(side.cuda_stream)), "mzk_srs_register_dev")
    d_bases.zero_()                                                   # the library keeps its own copy
    torch.cuda.synchronize()
    pp_dev = mj.UnivariateProverParam(c, h.value, n)
    pp_host = mj.UnivariateProverParam.from_affine(curve_id, bases)
    want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases, scalars, threads=8))[0]
    for off, m in ((0, n), (7, n - 7), (0, 1), (n - 1, 1), (100, 1500)):
        w = want if (off, m) == (0, n) else cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[off:off + m], scalars[:m], threads=8))[0]
        got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp_dev, scalars[:m], base_offset=off))[0]
        assert np.array_equal(got, w), (off, m)
        assert np.array_equal(cref.jac_to_affine(curve_id, mj.msm_bigint(pp_host, scalars[:m], base_offset=off))[0], w), (off, m)
    assert np.array_equal(pp_dev.powers_of_g(0, n), bases)            # mzk_srs_download gives the registered points back
    h0 = C.c_uint64()
    gpu.check(L.mzk_srs_register_dev(curve_id, None, 0, C.byref(h0), None), "mzk_srs_register_dev")
    empty = mj.UnivariateProverParam(c, h0.value, 0)
    jac = mj.msm_bigint(empty, scalars[:4])
    assert not np.any(jac[2]), "an MSM over an empty key is the point at infinity (Z = 0)"
    assert L.mzk_srs_register_dev(7, C.c_void_p(d_bases.data_ptr()), n, C.byref(h0), None) != 0       # unknown curve
    assert L.mzk_srs_register_dev(curve_id, None, 5, C.byref(h0), None) != 0                           # null pointer with points
    for p in (pp_dev, pp_host, empty):
        p.release()


@pytest.mark.parametrize("curve_id,log_n", [(0, 6), (1, 5), (0, 10)])
def test_quotient_combine_dev_is_the_inverse_of_the_class_remainders(gpu, mj, curve_id, log_n):
    """mzk_plonk_quotient_combine_dev (all eight residue classes of the 8n-point coset g H_8n, class-major): from the remainders
    t mod (X^n - h_k^n), h_k = g w_8n^k, computed with big integers from a random t of 8n coefficients, it must return t -- what
    `coset.ifft` returns at prover.rs:672."""
    import torch
    c = mj.params.CURVES[curve_id]
    r, n = c.r, 1 << log_n
    rng = random.Random(1000 + log_n)
    t = [rng.randrange(r) for _ in range(8 * n)]
    w = pow(c.fr_generator, (r - 1) // (8 * n), r)
    rem = []
    for k in range(8):
        hk_n = pow(c.fr_generator * pow(w, k, r) % r, n, r)
        pw = [pow(hk_n, q, r) for q in range(8)]
        rem.append([sum(t[j + q * n] * pw[q] for q in range(8)) % r for j in range(n)])
    d_rem = torch.from_numpy(mj.params.fr_to_mont(c, [v for row in rem for v in row]).view(np.int64)).cuda()
    d_out = torch.zeros((8 * n, 4), dtype=torch.int64, device="cuda")
    L = mj.load()
    st = torch.cuda.current_stream().cuda_stream
    gpu.check(L.mzk_plonk_quotient_combine_dev(curve_id, log_n, C.c_void_p(d_rem.data_ptr()), C.c_void_p(d_out.data_ptr()), C.c_void_p(st)), "mzk_plonk_quotient_combine_dev")
    torch.cuda.synchronize()
    got = mj.params.fr_from_mont(c, d_out.cpu().numpy().view(np.uint64))
    assert got == t
    assert L.mzk_plonk_quotient_combine_dev(curve_id, log_n, None, C.c_void_p(d_out.data_ptr()), None) != 0
    assert L.mzk_plonk_quotient_combine_dev(5, log_n, C.c_void_p(d_rem.data_ptr()), C.c_void_p(d_out.data_ptr()), None) != 0
