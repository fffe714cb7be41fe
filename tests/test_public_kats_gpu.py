"""GPU: the public doublings of tests/test_public_kats.py through the DEVICE -- the MSM (both paths) on the one pair (2, G), the
device generator of the testing SRS (gen_srs_for_testing with beta = 2: point 1 is [2]G), and a commitment to 2 X^0."""
import numpy as np
import pytest

from test_public_kats import PUBLIC_2G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve_id", [0, 1])
def test_device_msm_and_srs_generator_against_public_doublings(gpu, mj, curve_id):
    import pyref as P
    c, pc = mj.params.CURVES[curve_id], P.CURVES[curve_id]
    g = np.concatenate([mj.params.fq_to_mont(c, [pc.gx])[0], mj.params.fq_to_mont(c, [pc.gy])[0]]).reshape(1, -1)
    pp = mj.UnivariateProverParam.from_affine(curve_id, np.repeat(g, 2048, axis=0))
    two = mj.params.fr_bigints([2])
    L = mj.load()
    for table in (1, 0):
        L.mzk_msm_set_precompute(table)
        try:
            aff = mj.jacobian_to_affine(c, mj.msm_bigint(pp, two)[None])[0]
            assert tuple(mj.params.fq_from_mont(c, aff)) == PUBLIC_2G[curve_id], table
            # ... and as 2048 pairs (1, G) + (1, G) + 0 ...: the table path (n >= 1024) with two occupied buckets
            ones = np.zeros((2048, 4), dtype=np.uint64)
            ones[0, 0] = ones[1777, 0] = 1
            aff = mj.jacobian_to_affine(c, mj.msm_bigint(pp, ones)[None])[0]
            assert tuple(mj.params.fq_from_mont(c, aff)) == PUBLIC_2G[curve_id], table
        finally:
            L.mzk_msm_set_precompute(1)
    pp.release()
    srs = mj.UnivariateProverParam.gen_srs_for_testing(c, 2, 3)                    # [1]G, [2]G, [4]G, [8]G
    pts = srs.powers_of_g()
    assert tuple(mj.params.fq_from_mont(c, pts[0])) == (pc.gx, pc.gy)
    assert tuple(mj.params.fq_from_mont(c, pts[1])) == PUBLIC_2G[curve_id]
    com = mj.UnivariateKzgPCS.commit(srs, mj.params.fr_to_mont(c, [2]))             # the polynomial 2: [2]G again, through commit
    assert tuple(mj.params.fq_from_mont(c, com.xy)) == PUBLIC_2G[curve_id]
    srs.release()
