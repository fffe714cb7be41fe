"""GPU parity: the HIP Pippenger MSM / KZG commit (through the C ABI) against the committed
vectors, the C oracle on seeded inputs, and the trapdoor identity at the benchmark size."""
import random

import numpy as np
import pytest

from conftest import (affine_from_limbs, affine_limbs, fr_mont_limbs, golden_pt, jacobian_to_affine_ints, load_golden)

pytestmark = pytest.mark.gpu


def _bigints(scalars):
    return np.array([[(s >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for s in scalars], dtype=np.uint64).reshape(-1, 4)


def test_msm_matches_golden_vectors(gpu, mj, pyref):
    for case in load_golden("msm_vectors"):
        c = pyref.CURVES[case["curve"]]
        pp = mj.UnivariateProverParam.from_affine(c.curve_id, affine_limbs(c, [golden_pt(p) for p in case["bases"]]))
        scalars = [int(s, 16) for s in case["scalars"]]
        jac = mj.msm_bigint(pp, _bigints(scalars))
        assert jacobian_to_affine_ints(c, jac) == golden_pt(case["result"]), (c.name, len(scalars))
        if max(scalars) < c.r:
            jac = mj.msm_bigint(pp, fr_mont_limbs(c, scalars), scalars_are_mont=True)
            assert jacobian_to_affine_ints(c, jac) == golden_pt(case["result"]), (c.name, len(scalars), "mont")
        pp.release()


def test_commit_matches_golden_trapdoor_vectors(gpu, mj, pyref):
    for case in load_golden("kzg_vectors"):
        c = pyref.CURVES[case["curve"]]
        pp = mj.UnivariateProverParam.gen_srs_for_testing(c.curve_id, int(case["beta"], 16), len(case["srs"]) - 1)
        srs = pp.powers_of_g()
        for i, p in enumerate(case["srs"]):
            assert affine_from_limbs(c, srs[i]) == golden_pt(p), ("srs", c.name, i)
        com = mj.UnivariateKzgPCS.commit(pp, fr_mont_limbs(c, [int(v, 16) for v in case["coeffs"]]))
        assert affine_from_limbs(c, com.xy) == golden_pt(case["commitment"]), (c.name, len(case["coeffs"]))
        pp.release()


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 1000, 1 << 12, (1 << 14) + 3, 1 << 16])
def test_msm_matches_c_oracle(gpu, mj, cref, curve_id, n):
    c = mj.params.CURVES[curve_id]
    bases = cref.g1_arith_bases(curve_id, 0xabcdef12345 + n, 0x777, n)
    scalars = mj.params.random_fr_mont(c, n, seed=n)
    edge = [0, 1, 2, c.r - 1, (1 << 15) - 1, 1 << 15, (1 << 16) - 1, 1 << 16, (1 << 16) + 1, (1 << 240) - 1, c.r >> 1]
    for i, e in enumerate(edge):
        if i * 2 < n:
            scalars[i * 2] = _bigints([e])[0]
    if n >= 8:
        bases[7] = bases[6]                       # equal points with different scalars
    if n >= 40:
        bases[33] = 0                             # infinity in the base table
        scalars[35] = scalars[34]
        bases[35, 0] = bases[34, 0]               # P and -P under the same scalar
        neg = cref.g1_mul(curve_id, bases[34], c.r - 1)
        bases[35] = neg
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases, scalars, threads=8))[0]
    got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, scalars))[0]
    assert np.array_equal(got, want)
    # Montgomery-form scalars (what a DensePolynomial holds) give the same point
    got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, cref.fr_convert(curve_id, scalars, True), scalars_are_mont=True))[0]
    assert np.array_equal(got, want)
    # base_offset = num_leading_zeros: a suffix of the SRS
    if n > 5:
        want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[5:], scalars[5:], threads=8))[0]
        got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, scalars[5:], base_offset=5))[0]
        assert np.array_equal(got, want)
    pp.release()


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("table", [1, 0])
def test_bucket_reduction_exceptional_operands(gpu, mj, cref, curve_id, table):
    """The levels of the bucket reduction add BUCKETS to each other (msm.cuh fold_one / fold_one_quad, four lanes per addition on the
    narrow levels): with all bases equal and scalars 1 .. n every bucket holds the same point, so every addition of every level is a
    doubling; with the upper half of the bases negated the first level cancels to infinity and the later ones add infinities.  Both must
    take the exceptional path of xyzzx_add_quad and give the oracle's point."""
    from mpc_jellyfish_amd import lib as mlib
    c = mj.params.CURVES[curve_id]
    n = 4096
    g = cref.g1_arith_bases(curve_id, 5, 7, 1)[0]
    neg = cref.g1_mul(curve_id, g, c.r - 1)
    L = mlib.ensure_init()
    L.mzk_msm_set_precompute(table)
    try:
        for kind in ("doublings", "cancel", "mixed"):
            bases = np.repeat(g[None], n, axis=0)
            scalars = _bigints(list(range(1, n + 1)))
            if kind == "cancel":
                bases[n // 2:] = neg
            if kind == "mixed":                              # a few distinct multiples, signs by parity: equal, inverse and ordinary pairs side by side
                rng = np.random.default_rng(11)
                scalars = _bigints([int(v) for v in rng.integers(1, 64, size=n)])
                bases[1::2] = neg
            pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
            want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases, scalars, threads=8))[0]
            got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, scalars))[0]
            assert np.array_equal(got, want), kind
            pp.release()
    finally:
        L.mzk_msm_set_precompute(1)


def test_launch_count(gpu, mj, cref):
    """mzk_launch_count: kernels launched so far; an MSM adds a few dozen, a null pointer is refused."""
    import ctypes as C
    from mpc_jellyfish_amd import lib as mlib
    L = mlib.ensure_init()
    pp = mj.UnivariateProverParam.gen_srs_for_testing(0, 77, 2047)
    scalars = mj.params.random_fr_mont(mj.params.CURVES[0], 2048, seed=1)
    mj.msm_bigint(pp, scalars, scalars_are_mont=True)                    # (builds the fixed-base table)
    a, b = C.c_uint64(), C.c_uint64()
    mlib.check(L.mzk_launch_count(C.byref(a)), "mzk_launch_count")
    mj.msm_bigint(pp, scalars, scalars_are_mont=True)
    mlib.check(L.mzk_launch_count(C.byref(b)), "mzk_launch_count")
    assert 10 <= b.value - a.value <= 80
    assert L.mzk_launch_count(None) == -1                                # MZK_ERR_INVALID_ARG
    pp.release()


def test_msm_skewed_and_degenerate_scalars(gpu, mj, cref):
    """All points in one bucket per window; all-zero scalars; a single non-zero scalar."""
    curve_id, n = 0, 5000
    c = mj.params.CURVES[curve_id]
    bases = cref.g1_arith_bases(curve_id, 99, 3, n)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    same = np.repeat(_bigints([0x1234567890abcdef1234567890abcdef1234567890abcdef]), n, axis=0)
    want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases, same, threads=8))[0]
    assert np.array_equal(cref.jac_to_affine(curve_id, mj.msm_bigint(pp, same))[0], want)
    zeros = np.zeros((n, 4), dtype=np.uint64)
    assert not cref.jac_to_affine(curve_id, mj.msm_bigint(pp, zeros))[0].any()
    assert not mj.msm_bigint(pp, zeros[:0])[2].any()                     # n = 0 -> Z = 0
    one = zeros.copy()
    one[n - 1] = _bigints([c.r - 1])[0]
    want = cref.g1_mul(curve_id, bases[n - 1], c.r - 1)
    assert np.array_equal(cref.jac_to_affine(curve_id, mj.msm_bigint(pp, one))[0], want)
    pp.release()


@pytest.mark.parametrize("mode", ["all_equal", "half_equal", "small", "two_values"])
def test_msm_skewed_large(gpu, mj, cref, mode):
    """Adversarial scalar mixes at 2^18: over-long buckets take the chunked path (msm_long_* kernels);
    small scalars leave the high windows empty (the bench circuit's witness is 0..2^20)."""
    import time
    curve_id, n = 0, 1 << 18
    c = mj.params.CURVES[curve_id]
    bases = cref.g1_arith_bases(curve_id, 31337, 11, n)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    rnd = mj.params.random_fr_mont(c, n, seed=5)
    if mode == "all_equal":
        scalars = np.repeat(_bigints([0x5a5a5a5a1234567890abcdef0fedcba987654321deadbeefcafef00d12345678 % c.r]), n, axis=0)
    elif mode == "half_equal":
        scalars = rnd.copy()
        scalars[::2] = _bigints([c.r - 2])[0]
    elif mode == "small":
        scalars = np.zeros((n, 4), dtype=np.uint64)
        scalars[:, 0] = np.arange(n, dtype=np.uint64)
    else:
        scalars = np.where((np.arange(n) % 3 == 0)[:, None], _bigints([(1 << 200) + 12345])[0], _bigints([c.r >> 3])[0]).astype(np.uint64)
    t0 = time.time()
    got = mj.msm_bigint(pp, scalars)
    assert time.time() - t0 < 5.0, "skewed MSM fell off a performance cliff"
    want = cref.msm(curve_id, bases, scalars, threads=8)
    assert np.array_equal(cref.jac_to_affine(curve_id, got)[0], cref.jac_to_affine(curve_id, want)[0])
    pp.release()


def test_commit_api_behaviour(gpu, mj, cref):
    """Degree guard (mod.rs:98-104), leading/trailing zero handling, batch_commit, device scalars."""
    import torch
    curve_id = 1
    c = mj.params.CURVES[curve_id]
    pp_full = mj.UnivariateProverParam.gen_srs_for_testing(curve_id, 0xdeadbeef, 40)
    pp = pp_full.trim(20)                                                  # 21 powers
    srs = pp_full.powers_of_g()
    assert cref.count_off_curve(curve_id, srs) == 0
    polys = [mj.params.random_fr_mont(c, k, seed=k) for k in (1, 7, 21)]
    polys[1][:3] = 0
    coms = mj.UnivariateKzgPCS.batch_commit(pp, polys)
    for p, com in zip(polys, coms):
        want = cref.jac_to_affine(curve_id, cref.msm(curve_id, srs[:len(p)], p, scalars_are_mont=True))[0]
        assert np.array_equal(com.xy, want)
    too_long = np.zeros((23, 4), dtype=np.uint64)
    too_long[22] = 1
    with pytest.raises(mj.PCSError):
        mj.UnivariateKzgPCS.commit(pp, too_long)                           # degree 22 > 21
    assert mj.UnivariateKzgPCS.commit(pp, np.zeros((5, 4), dtype=np.uint64)).is_infinity()
    with pytest.raises(mj.PCSError):
        pp.trim(30)
    with pytest.raises(mj.MzkError):
        mj.msm_bigint(mj.UnivariateProverParam(c, 999999, 4, owner=False), polys[1])   # unknown handle
    t = torch.from_numpy(polys[2].view(np.int64)).cuda()
    jac = mj.msm_bigint(pp, t, scalars_are_mont=True)
    assert np.array_equal(cref.jac_to_affine(curve_id, jac)[0], coms[2].xy)
    pp_full.release()


@pytest.mark.parametrize("curve_id,log_n", [(0, 20), (1, 22), (0, 24)])
def test_msm_full_size_trapdoor(gpu, mj, cref, curve_id, log_n):
    """BASELINE config C2 (2^20 pairs, BLS12-381), C5's commitment size (2^22, BN254) and a 2^24-pair MSM (a 24 GB table of
    precomputed multiples): commit(p) over [beta^i]G equals [p(beta)]G -- one oracle scalar multiplication pins an MSM of any
    size."""
    import torch
    n = 1 << log_n
    c = mj.params.CURVES[curve_id]
    beta = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8091a2b3c4d5e6f708192a3b4c5d6e7f % c.r
    pp = mj.UnivariateProverParam.gen_srs_for_testing(curve_id, beta, n - 1)
    srs_sample = pp.powers_of_g(n - 3, 3)
    assert cref.count_off_curve(curve_id, srs_sample) == 0
    coeffs = mj.params.random_fr_mont(c, n, seed=2020)
    t = torch.from_numpy(coeffs.view(np.int64)).cuda()
    jac = mj.msm_bigint(pp, t, scalars_are_mont=True)
    p_beta = cref.poly_eval(curve_id, coeffs, mj.params.fr_to_mont(c, [beta])[0])
    k = mj.params.limbs_to_int(cref.fr_convert(curve_id, p_beta.reshape(1, 4), False)[0])
    assert np.array_equal(cref.jac_to_affine(curve_id, jac)[0], cref.g1_mul_gen(curve_id, k))
    # linearity: MSM(2x) = 2 MSM(x)
    doubled = cref.fr_mul(curve_id, coeffs, np.repeat(mj.params.fr_to_mont(c, [2]), n, axis=0))
    jac2 = mj.msm_bigint(pp, doubled, scalars_are_mont=True)
    assert np.array_equal(cref.jac_to_affine(curve_id, jac2)[0], cref.g1_mul_gen(curve_id, 2 * k % c.r))
    pp.release()


@pytest.mark.parametrize("mode", ["one_bucket", "u8", "booleans", "iota", "repeats"])
def test_msm_heavy_buckets_full_size_trapdoor(gpu, mj, cref, mode):
    """Scalar distributions that put most of 2^20 x 13 digits into a handful of buckets -- what witness VALUES look like (small
    numbers, flags, counters) -- against the trapdoor identity commit(p) = [p(beta)]G.  They take the msm_heavy_* kernels (workgroup
    per run of a heavy bucket, workgroup trees over the runs; `one_bucket` -- every scalar equal to d (1 + 2^20 + 2^40 + ..), all 13
    digits of all scalars in ONE bucket -- needs all three tree levels) and the per-run msm_long_combine_kernel (`repeats`)."""
    import time
    import torch
    curve_id, n = 0, 1 << 20
    c = mj.params.CURVES[curve_id]
    beta = 0x2b3c4d5e6f708192a3b4c5d6e7f8091a2b3c4d5e6f708192a3b4c5d6e7f80912 % c.r
    pp = mj.UnivariateProverParam.gen_srs_for_testing(curve_id, beta, n - 1)
    rng = np.random.default_rng(99)
    if mode == "one_bucket":
        d = 0x2f0f1
        vals = np.repeat(_bigints([sum(d << (20 * i) for i in range(12)) + (5 << 240)]), n, axis=0)
    elif mode == "repeats":
        few = [int.from_bytes(rng.bytes(32), "little") % c.r for _ in range(4096)]
        vals = np.tile(_bigints(few), (n // 4096, 1))
    else:
        vals = np.zeros((n, 4), dtype=np.uint64)
        vals[:, 0] = {"u8": rng.integers(0, 256, n), "booleans": rng.integers(0, 2, n), "iota": np.arange(n)}[mode].astype(np.uint64)
    mont = cref.fr_convert(curve_id, vals, True)                          # canonical -> Montgomery
    t = torch.from_numpy(vals.view(np.int64)).cuda()
    mj.msm_bigint(pp, t)                                                  # (builds the table)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    jac = mj.msm_bigint(pp, t)
    ms = (time.perf_counter() - t0) * 1e3
    p_beta = cref.poly_eval(curve_id, mont, mj.params.fr_to_mont(c, [beta])[0])
    k = mj.params.limbs_to_int(cref.fr_convert(curve_id, p_beta.reshape(1, 4), False)[0])
    assert np.array_equal(cref.jac_to_affine(curve_id, jac)[0], cref.g1_mul_gen(curve_id, k))
    assert ms < 40.0, "heavy-bucket MSM fell off a performance cliff: %.1f ms" % ms
    pp.release()


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("log_n", [0, 3, 6, 10])
def test_lagrange_srs_for_testing_against_the_definition(gpu, mj, pyref, curve_id, log_n):
    """mzk_srs_generate_lagrange_for_testing: point i = [L_i(beta)]G with L_i(beta) = w^i (beta^n - 1) / (n (beta - w^i)) (big ints, one oracle
    scalar multiplication per sampled point), the extra points [beta^j (beta^n - 1)]G; sum_i points = G (the L_i sum to one); beta ON the
    domain gives the unit vector."""
    c = mj.params.CURVES[curve_id]
    pc = pyref.CURVES[curve_id]
    r, n = c.r, 1 << log_n
    w = pow(c.fr_generator, (r - 1) >> log_n, r) if log_n else 1
    G = pyref.g1_gen(pc)
    for beta in (0x1234567890abcdef1122334455667788 % r, pow(w, 5 % n, r)):
        pp = mj.UnivariateProverParam.gen_lagrange_srs_for_testing(c, beta, n, n_extra=3)
        pts = pp.powers_of_g()
        vanish = (pow(beta, n, r) - 1) % r
        for i in sorted({0, 1 % n, 5 % n, n // 2, n - 1}):
            wi = pow(w, i, r)
            li = (1 if beta == wi else 0) if vanish == 0 else wi * vanish % r * pow(n * (beta - wi) % r, -1, r) % r
            want = pyref.g1_mul(pc, li, G)
            assert affine_from_limbs(c, pts[i]) == want, (log_n, i)
        for j in range(3):
            want = pyref.g1_mul(pc, pow(beta, j, r) * vanish % r, G)
            assert affine_from_limbs(c, pts[n + j]) == want, ("extra", j)
        ones = np.repeat(_bigints([1]), n, axis=0)
        assert jacobian_to_affine_ints(pc, mj.msm_bigint(pp, ones)) == G
        pp.release()


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("log_n", [0, 1, 3, 7, 10])
def test_lagrange_key_from_the_points_of_an_srs(gpu, mj, curve_id, log_n):
    """mzk_srs_lagrange_from_srs (no trapdoor: the inverse NTT over the group of the SRS's first n points, then S_(n+j) - S_j) gives the
    very points mzk_srs_generate_lagrange_for_testing computes from beta -- standard generator and a base point of the caller's."""
    c = mj.params.CURVES[curve_id]
    n = 1 << log_n
    beta = 0x5eed0123456789abcdef00112233445566778899 % c.r
    for g in (None, "rand"):
        if g == "rand":
            g = mj.rng.g1_rand(c, mj.rng.test_rng())
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, beta, n + 2, g=g)
        want = mj.UnivariateProverParam.gen_lagrange_srs_for_testing(c, beta, n, n_extra=3, g=g)
        got = ck.lagrange_key(n, n_extra=3)
        assert np.array_equal(got.powers_of_g(), want.powers_of_g()), (log_n, g is None)
        for p in (ck, want, got):
            p.release()


@pytest.mark.parametrize("curve_id", [0, 1])
def test_lagrange_key_from_the_points_of_an_srs_at_2p16(gpu, mj, curve_id):
    """The same at 2^16 points (three passes of the group NTT's stage loop; 0.14 s): every point against the key computed from beta."""
    c = mj.params.CURVES[curve_id]
    n = 1 << 16
    beta = 0x0f1e2d3c4b5a69788796a5b4c3d2e1f00112233445566778 % c.r
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, beta, n + 2)
    want = mj.UnivariateProverParam.gen_lagrange_srs_for_testing(c, beta, n, n_extra=3)
    got = ck.lagrange_key(n, n_extra=3)
    assert np.array_equal(got.powers_of_g(), want.powers_of_g())
    for p in (ck, want, got):
        p.release()


def test_msm_batch_fused(gpu, mj, cref):
    """mzk_msm_batch / mzk_msm_batch_dev: MSMs of different lengths (two window sizes, an empty one)
    in one call equal the single calls and the oracle."""
    import torch
    curve_id = 0
    c = mj.params.CURVES[curve_id]
    nmax = (1 << 13) + 3
    bases = cref.g1_arith_bases(curve_id, 777, 5, nmax)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    lens = [nmax, 1 << 13, 0, 100, (1 << 13) + 2, 7, 1 << 12]
    offs = [0, 3, 0, 50, 1, 0, 9]
    sets = [mj.params.random_fr_mont(c, n, seed=40 + i) for i, n in enumerate(lens)]
    jac = mj.msm_bigint_batch(pp, sets, offs, scalars_are_mont=True)
    aff = mj.jacobian_to_affine(c, jac)
    for i, (n, o) in enumerate(zip(lens, offs)):
        want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[o:o + n], sets[i], scalars_are_mont=True, threads=8))[0]
        assert np.array_equal(aff[i], want), i
        assert np.array_equal(cref.jac_to_affine(curve_id, jac[i])[0], want), i
    dev_sets = [torch.from_numpy(s.view(np.int64)).cuda() for s in sets]
    jac2 = mj.msm_bigint_batch(pp, dev_sets, offs, scalars_are_mont=True)
    assert np.array_equal(mj.jacobian_to_affine(c, jac2), aff)
    pp.release()


@pytest.mark.parametrize("curve_id", [0, 1])
def test_msm_precomputed_table_path(gpu, mj, cref, curve_id):
    """n >= 2^10 runs on the precomputed-multiples table (one bucket set for all windows): sub-ranges of the SRS (base_offset), a batch
    mixing table and plain paths, and the switch that turns the table off must all give the oracle's point.  Both curves (BN254 Fq on
    9 x 29-bit limbs with the tight subtraction pads: tools/ecx_bounds.py)."""
    c = mj.params.CURVES[curve_id]
    n_srs = (1 << 17) + 64
    bases = cref.g1_arith_bases(curve_id, 0xfeed, 0x1d, n_srs)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    scalars = mj.params.random_fr_mont(c, n_srs, seed=171)
    scalars[5] = 0
    scalars[6] = _bigints([c.r - 1])[0]
    cases = [(0, n_srs), (7, 1 << 17), (64, 1 << 17), (3, (1 << 17) + 11)]
    for off, n in cases:
        want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[off:off + n], scalars[:n], threads=8))[0]
        got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, scalars[:n], base_offset=off))[0]
        assert np.array_equal(got, want), (off, n)
    import ctypes as C
    pts, tab = C.c_uint64(), C.c_uint64()
    mj.lib.check(mj.load().mzk_srs_hbm_bytes(pp.handle, C.byref(pts), C.byref(tab)), "mzk_srs_hbm_bytes")
    assert tab.value > 0 and mj.lib.msm_last_shape()[2] == 1 << (mj.lib.msm_last_shape()[0] - 1), "large MSMs must have taken the precomputed-table path"
    # batch: large (table) and small (plain) members interleaved
    sets = [scalars[:1 << 17], scalars[:1000], scalars[:(1 << 17) + 3], scalars[:0]]
    offs = [1, 2, 0, 0]
    jac = mj.msm_bigint_batch(pp, sets, offs)
    for i, (s, o) in enumerate(zip(sets, offs)):
        want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[o:o + len(s)], s, threads=8))[0]
        assert np.array_equal(cref.jac_to_affine(curve_id, jac[i])[0], want), i
    # same result with the table disabled
    L = mj.load()
    L.mzk_msm_set_precompute(0)
    try:
        got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, scalars[:1 << 17], base_offset=7))[0]
        assert mj.lib.msm_last_shape()[0] <= 16
    finally:
        L.mzk_msm_set_precompute(1)
    want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[7:7 + (1 << 17)], scalars[:1 << 17], threads=8))[0]
    assert np.array_equal(got, want)
    pp.release()


@pytest.mark.parametrize("curve_id", [0, 1])
def test_kzg_open_matches_definition(gpu, mj, pyref, curve_id):
    """UnivariateKzgPCS::open / batch_open (mod.rs:135-190): proof = [q(beta)]G for q = (p - p(z)) / (X - z), evaluation = p(z);
    through the trapdoor the verifier's pairing check reads p(beta) - p(z) == (beta - z) q(beta)."""
    c = pyref.CURVES[curve_id]
    r = c.r
    rng = random.Random(77 + curve_id)
    beta = rng.randrange(r)
    pp = mj.UnivariateProverParam.gen_srs_for_testing(c.curve_id, beta, 40)
    G = pyref.g1_gen(c)
    polys = [[rng.randrange(r) for _ in range(n)] for n in (41, 17, 2, 1)]
    polys[1][0] = polys[1][1] = 0                                        # low-order zero coefficients
    points = [rng.randrange(r) for _ in polys]
    points[2] = 0                                                        # division by X
    proofs, evals = mj.UnivariateKzgPCS.batch_open(pp, [fr_mont_limbs(c, p) for p in polys], points)
    for p, z, proof, ev in zip(polys, points, proofs, evals):
        assert ev == pyref.poly_eval(c, p, z)
        q_at_beta = (pyref.poly_eval(c, p, beta) - ev) * pow(beta - z, -1, r) % r
        want = pyref.g1_mul(c, q_at_beta, G) if q_at_beta else None
        got = None if proof.is_infinity() else affine_from_limbs(c, proof.xy)
        assert got == want
    with pytest.raises(mj.PCSError):
        mj.UnivariateKzgPCS.batch_open(pp, [fr_mont_limbs(c, polys[0])], points[:2])
    pp.release()


def test_concurrent_callers(gpu, mj, cref):
    """The reference calls msm_bigint and fft from Rayon workers at the same time (univariate_kzg/mod.rs:125-127,
    snark.rs:562-571).  Eight threads hammer mzk_msm / mzk_ntt / mzk_msm_batch concurrently (ctypes releases the GIL):
    every result must equal the single-threaded one."""
    import threading
    curve_id = 0
    c = mj.params.CURVES[curve_id]
    n = 1 << 12
    bases = cref.g1_arith_bases(curve_id, 5150, 9, n)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    dom = mj.Radix2EvaluationDomain(c, 12)
    sets = [mj.params.random_fr_mont(c, n, seed=900 + i) for i in range(8)]
    aff = lambda jac: mj.jacobian_to_affine(c, jac)          # the Jacobian representative depends on the (atomic) bucket order; the point does not
    want_msm = [aff(mj.msm_bigint(pp, s, scalars_are_mont=True))[0] for s in sets]
    want_ntt = [dom.fft(s) for s in sets]
    errors = []

    def worker(i):
        try:
            for rep in range(6):
                k = (i + rep) % 8
                if rep % 3 == 0:
                    got = aff(mj.msm_bigint(pp, sets[k], scalars_are_mont=True))[0]
                    assert np.array_equal(got, want_msm[k])
                elif rep % 3 == 1:
                    assert np.array_equal(dom.fft(sets[k]), want_ntt[k])
                else:
                    got = aff(mj.msm_bigint_batch(pp, [sets[k], sets[(k + 1) % 8]], scalars_are_mont=True))
                    assert np.array_equal(got[0], want_msm[k]) and np.array_equal(got[1], want_msm[(k + 1) % 8])
        except Exception as e:                                           # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    pp.release()


def test_error_paths(gpu, mj):
    """Bad handles, null pointers and sizes beyond the SRS come back as error codes with a message -- never a crash
    (the boundary must not unwind or abort: SURVEY.md 8(b))."""
    import ctypes as C
    L = mj.load()
    out = np.zeros(18, dtype=np.uint64)
    sc = np.zeros((4, 4), dtype=np.uint64)
    assert L.mzk_msm(0xdead, 0, sc.ctypes.data_as(C.c_void_p), 4, 0, out.ctypes.data_as(C.c_void_p)) == -4          # MZK_ERR_BAD_HANDLE
    assert b"handle" in L.mzk_last_error()
    pp = mj.UnivariateProverParam.gen_srs_for_testing(0, 5, 7)                                                        # 8 points
    assert L.mzk_msm(pp.handle, 0, sc.ctypes.data_as(C.c_void_p), 9, 0, out.ctypes.data_as(C.c_void_p)) == -1       # longer than the SRS
    assert L.mzk_msm(pp.handle, 6, sc.ctypes.data_as(C.c_void_p), 3, 0, out.ctypes.data_as(C.c_void_p)) == -1       # offset + n beyond it
    assert L.mzk_msm(pp.handle, 0, None, 4, 0, out.ctypes.data_as(C.c_void_p)) == -1                                 # null scalars
    assert L.mzk_ntt(7, sc.ctypes.data_as(C.c_void_p), 4, 2, 0, None) == -1                                          # unknown curve
    assert L.mzk_ntt(1, sc.ctypes.data_as(C.c_void_p), 4, 29, 0, None) == -1                                         # beyond BN254's two-adicity
    assert L.mzk_plonk_pk_release(12345) == -4
    assert L.mzk_strerror(-8).startswith(b"Plookup")
    # round-3 entry points: the Lagrange-basis key needs 2^log_n + n_extra points in the SRS; the quotient's top coefficients need a chunked
    # proving key; the division with remainder needs somewhere to put it
    h = C.c_uint64()
    assert L.mzk_srs_lagrange_from_srs(pp.handle, 3, 3, C.byref(h)) == -1                                            # 8 + 3 > 8 points
    assert L.mzk_srs_lagrange_from_srs(0xdead, 2, 1, C.byref(h)) == -4
    assert L.mzk_srs_lagrange_from_srs(pp.handle, 2, 3, C.byref(h)) == 0 and L.mzk_srs_release(h.value) == 0
    beta = np.array([5, 0, 0, 0], dtype=np.uint64)
    assert L.mzk_srs_generate_lagrange_for_testing(0, beta.ctypes.data_as(C.c_void_p), None, 40, 3, C.byref(h)) == -1
    assert L.mzk_srs_generate_lagrange_for_testing(0, beta.ctypes.data_as(C.c_void_p), None, 4, 99, C.byref(h)) == -1
    assert L.mzk_plonk_quotient_top_dev(12345, None, 0, 0, None, None, None, None, None, None) == -4
    assert L.mzk_poly_div_linear_rem_dev(0, None, 4, beta.ctypes.data_as(C.c_void_p), None, None, None) == -1
    released = pp.handle
    pp.release()
    assert L.mzk_srs_release(released) == -4, "releasing a handle twice is an error, not a crash"              # MZK_ERR_BAD_HANDLE
    assert L.mzk_msm(released, 0, sc.ctypes.data_as(C.c_void_p), 4, 0, out.ctypes.data_as(C.c_void_p)) == -4        # and so is using it


@pytest.mark.parametrize("curve_id", [0, 1])
def test_msm_random_ragged_sweep(gpu, mj, cref, curve_id):
    """One SRS, forty random (length, base_offset, scalar shape) MSMs on it -- lengths on both sides of the precomputed-table
    threshold (1024) and of the bucket-splitting rules, scalars uniform / tiny (witness-like) / sparse / all equal / top-heavy --
    each against the C restatement of ark-ec's Pippenger."""
    c = mj.params.CURVES[curve_id]
    rng = random.Random(4242 + curve_id)
    N = 6000
    bases = cref.g1_arith_bases(curve_id, 0x5eed + curve_id, 0x1234567, N)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    lengths = [1, 2, 1023, 1024, 1025, 2047, 2049, 4095, 4097, N] + [rng.randrange(1, N) for _ in range(30)]
    for case, n in enumerate(lengths):
        off = rng.randrange(0, N - n + 1)
        shape = case % 5
        if shape == 0:
            ints = [rng.randrange(c.r) for _ in range(n)]
        elif shape == 1:
            ints = [rng.randrange(1 << 20) for _ in range(n)]                       # wire values of a small circuit
        elif shape == 2:
            ints = [rng.randrange(c.r) if rng.random() < 0.05 else 0 for _ in range(n)]
        elif shape == 3:
            ints = [rng.randrange(c.r)] * n
        else:
            ints = [(c.r - 1 - rng.randrange(1 << 12)) for _ in range(n)]           # every top digit at its maximum
        scalars = _bigints(ints)
        want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[off:off + n], scalars, threads=8))[0]
        got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, scalars, base_offset=off))[0]
        assert np.array_equal(got, want), (case, n, off, shape)
    pp.release()


@pytest.mark.parametrize("curve_id", [0, 1])
def test_srs_precompute_report_and_same_points(gpu, mj, cref, curve_id):
    """mzk_srs_precompute builds the fixed-base table at once and reports its size; MSMs with the table and with it switched off
    (mzk_msm_set_precompute) return the same group element."""
    import ctypes as C
    L = mj.load()
    c = mj.params.CURVES[curve_id]
    n = 3000
    bases = cref.g1_arith_bases(curve_id, 0x1d + curve_id, 0x99, n)
    pp = mj.UnivariateProverParam.from_affine(curve_id, bases)
    bits, levels, nbytes, ms = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_double()
    assert L.mzk_srs_precompute(pp.handle, C.byref(bits), C.byref(levels), C.byref(nbytes), C.byref(ms)) == 0
    assert bits.value in (15, 16, 17, 20) and levels.value == (256 + bits.value) // bits.value      # signed digits of a 256-bit integer
    assert nbytes.value == levels.value * n * (112 if curve_id == 0 else 72) and ms.value > 0            # 2 x 14 / 2 x 9 limbs of 29 bits
    first = ms.value
    assert L.mzk_srs_precompute(pp.handle, None, None, None, C.byref(ms)) == 0 and ms.value == first, "built once"
    assert L.mzk_srs_precompute(0xdead, None, None, None, None) == -4
    sc = mj.params.random_fr_mont(c, n, seed=8)
    with_table = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, sc, scalars_are_mont=True))[0]
    L.mzk_msm_set_precompute(0)
    try:
        without = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, sc, scalars_are_mont=True))[0]
    finally:
        L.mzk_msm_set_precompute(1)
    want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases, sc, scalars_are_mont=True, threads=4))[0]
    assert np.array_equal(with_table, want) and np.array_equal(without, want)
    pp.release()
