"""GPU: sparse and skewed scalar vectors through both MSM paths at sizes where the round-3 changes apply -- the occupancy flags of
the bucket reduction (a bucket that holds nothing is neither written nor folded: the reference's bench circuit commits two zero
wire polynomials, plonk/benches/bench.rs:29-46 through univariate_kzg/mod.rs:90-131) and the two-level sort of large plain-path
MSMs (the variable-base path: `VariableBaseMSM::msm_bigint`, mod.rs:109-111).  Every result against the C oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _patterns(mj, c, n):
    from conftest import fr_mont_limbs
    rs = np.random.default_rng(5)
    dense = mj.params.random_fr_mont(c, n, seed=41)
    zero = np.zeros((n, 4), dtype=np.uint64)
    few = zero.copy()
    for i in (0, 1, n // 2, n - 1):                                   # what a masked zero polynomial looks like: a handful of coefficients
        few[i] = dense[i]
    one_val = np.repeat(dense[3:4], n, axis=0)                        # all equal: every point of a window in ONE bucket (the long-bucket path)
    plus_minus = one_val.copy()                                       # s, -s, s, -s ...: runs that cancel to infinity inside a bucket
    neg = fr_mont_limbs(c, [(-int(x)) % c.r for x in [mj.params.fr_from_mont(c, dense[3:4])[0]]])[0]
    plus_minus[1::2] = neg
    small = zero.copy()
    small[:, 0] = rs.integers(0, 1 << 16, size=n, dtype=np.uint64)    # 16-bit scalars in Montgomery *limbs*: arbitrary values < r, fine
    half = dense.copy()
    half[::2] = 0
    return {"dense": dense, "zero": zero, "few": few, "all_equal": one_val, "plus_minus": plus_minus, "small_limbs": small, "half": half}


@pytest.mark.parametrize("curve_id", [0, 1])
def test_sparse_and_skewed_scalars_both_paths(gpu, mj, cref, curve_id):
    c = mj.params.CURVES[curve_id]
    n = (1 << 17) + 5
    bases = cref.g1_arith_bases(curve_id, 0xbeef, 0x2b, n)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    pats = _patterns(mj, c, n)
    want = {k: cref.jac_to_affine(curve_id, cref.msm(curve_id, bases, v, scalars_are_mont=True, threads=8))[0] for k, v in pats.items()}
    L = mj.load()
    for table in (1, 0):
        L.mzk_msm_set_precompute(table)
        try:
            for k, v in pats.items():
                got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, v, scalars_are_mont=True))[0]
                assert np.array_equal(got, want[k]), (curve_id, table, k)
            assert (mj.lib.msm_last_shape()[0] > 16) == bool(table)
            # one fused batch: dense and (nearly) empty members share one bucket reduction
            names = ["dense", "few", "zero", "half", "zero", "all_equal", "plus_minus"]
            jac = mj.msm_bigint_batch(pp, [pats[k] for k in names], scalars_are_mont=True)
            for i, k in enumerate(names):
                assert np.array_equal(cref.jac_to_affine(curve_id, jac[i])[0], want[k]), (curve_id, table, "batch", k)
        finally:
            L.mzk_msm_set_precompute(1)
    pp.release()


def test_plain_path_two_level_sort_at_full_size(gpu, mj):
    """2^20 pairs with the table off (bench.py's headline shape: 16 windows of 2^15 buckets) against the table path and the trapdoor:
    commit(p) over [beta^i]G is [p(beta)]G."""
    import torch
    c = mj.params.BLS12_381
    n = 1 << 20
    beta = 0x5eed1234567
    pp = mj.UnivariateProverParam.gen_srs_for_testing(c, beta, n - 1)
    x = mj.params.random_fr_mont(c, n, seed=9)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    L = mj.load()
    with_table = mj.jacobian_to_affine(c, mj.msm_bigint(pp, d, scalars_are_mont=True)[None])[0]
    L.mzk_msm_set_precompute(0)
    try:
        plain = mj.jacobian_to_affine(c, mj.msm_bigint(pp, d, scalars_are_mont=True)[None])[0]
        assert mj.lib.msm_last_shape() == (16, 16, 1 << 15)
    finally:
        L.mzk_msm_set_precompute(1)
    assert np.array_equal(plain, with_table)
    # p(beta) by Horner on the host, then ONE scalar multiplication: an MSM of one pair on the same SRS
    vals = mj.params.fr_from_mont(c, x)
    acc = 0
    for v in reversed(vals):
        acc = (acc * beta + int(v)) % c.r
    from conftest import fr_mont_limbs
    one = mj.jacobian_to_affine(c, mj.msm_bigint(pp, fr_mont_limbs(c, [acc]), scalars_are_mont=True)[None])[0]
    assert np.array_equal(plain, one)
    pp.release()
