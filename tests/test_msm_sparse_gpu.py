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
    shapes = {}
    for table in (1, 0):
        L.mzk_msm_set_precompute(table)
        try:
            for k, v in pats.items():
                got = cref.jac_to_affine(curve_id, mj.msm_bigint(pp, v, scalars_are_mont=True))[0]
                assert np.array_equal(got, want[k]), (curve_id, table, k)
            shapes[table] = mj.lib.msm_last_shape()                      # (window bits, digits per scalar, buckets per set)
            # one fused batch: dense and (nearly) empty members share one bucket reduction
            names = ["dense", "few", "zero", "half", "zero", "all_equal", "plus_minus"]
            jac = mj.msm_bigint_batch(pp, [pats[k] for k in names], scalars_are_mont=True)
            for i, k in enumerate(names):
                assert np.array_equal(cref.jac_to_affine(curve_id, jac[i])[0], want[k]), (curve_id, table, "batch", k)
        finally:
            L.mzk_msm_set_precompute(1)
    assert shapes[1][0] > shapes[0][0], "the table path (window 16 / 17 at this SRS size) and the plain path (window log2 n - 2) must both have run"
    pp.release()


def test_plain_path_two_level_sort_at_full_size(gpu, mj, cref):
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
    # ... and against the ORACLE's scalar multiplication [p(beta)]G (oracle/cpu_ref.c), not only the library's own 1-pair MSM
    assert np.array_equal(plain, cref.g1_mul_gen(0, acc))
    pp.release()


def test_large_sorts_keep_to_the_regular_path_at_2p22(gpu, mj, cref):
    """2^22 pairs: the fine bins of the two-level sort are halved until they hold about 64 K records (msm.hip; at 2^11 buckets per bin every
    bin of a 2^22-pair sort was 'huge').  Table path and plain path (different bin layouts: two regions / uniform) against each other and
    against the oracle's [p(beta)]G."""
    import torch
    c = mj.params.BLS12_381
    n = 1 << 22
    beta = 0x7e57ab1e5
    pp = mj.UnivariateProverParam.gen_srs_for_testing(c, beta, n - 1)
    x = mj.params.random_fr_mont(c, n, seed=22)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    L = mj.load()
    with_table = mj.jacobian_to_affine(c, mj.msm_bigint(pp, d, scalars_are_mont=True)[None])[0]
    assert mj.lib.msm_last_shape()[0] == 20
    L.mzk_msm_set_precompute(0)
    try:
        plain = mj.jacobian_to_affine(c, mj.msm_bigint(pp, d, scalars_are_mont=True)[None])[0]
        assert mj.lib.msm_last_shape() == (16, 16, 1 << 15)
    finally:
        L.mzk_msm_set_precompute(1)
    assert np.array_equal(plain, with_table)
    acc = 0
    for v in reversed(mj.params.fr_from_mont(c, x)):
        acc = (acc * beta + int(v)) % c.r
    assert np.array_equal(plain, cref.g1_mul_gen(0, acc))
    pp.release()


def test_srs_whose_table_does_not_fit_commits_on_the_plain_path(gpu, mj, cref):
    """MZK_MSM_TABLE_BUDGET (bytes): an SRS whose fixed-base table would exceed it gets none -- mzk_srs_precompute reports zeros -- and its
    commitments come from the variable-base path: the oracle's point all the same (include/mzk.h, mzk_srs_precompute)."""
    import ctypes as C
    import os
    import torch
    c = mj.params.BLS12_381
    n = 1 << 14
    beta = 0xabcdef0123456789
    x = mj.params.random_fr_mont(c, n, seed=41)
    acc = 0
    for v in reversed(mj.params.fr_from_mont(c, x)):
        acc = (acc * beta + int(v)) % c.r
    want = cref.g1_mul_gen(0, acc)
    L = mj.load()
    os.environ["MZK_MSM_TABLE_BUDGET"] = str(1 << 20)                  # 1 MB: the 16 x 2^14 x 112 B table (29 MB) does not fit
    try:
        pp = mj.UnivariateProverParam.gen_srs_for_testing(c, beta, n - 1)
        bits, levels, nbytes, ms = C.c_uint32(7), C.c_uint32(7), C.c_uint64(7), C.c_double(7)
        mj.lib.check(L.mzk_srs_precompute(pp.handle, C.byref(bits), C.byref(levels), C.byref(nbytes), C.byref(ms)), "mzk_srs_precompute")
        assert (bits.value, levels.value, nbytes.value) == (0, 0, 0)
        pts, tab = C.c_uint64(), C.c_uint64()
        mj.lib.check(L.mzk_srs_hbm_bytes(pp.handle, C.byref(pts), C.byref(tab)), "mzk_srs_hbm_bytes")
        assert tab.value == 0 and pts.value == n * (96 + 112)
        d = torch.from_numpy(x.view(np.int64)).cuda()
        got = mj.jacobian_to_affine(c, mj.msm_bigint(pp, d, scalars_are_mont=True)[None])[0]
        assert mj.lib.msm_last_shape()[1] > 1                          # several windows with their own buckets: the plain path
        assert np.array_equal(got, want)
        batch = mj.jacobian_to_affine(c, mj.msm_bigint_batch(pp, [d, d[: n // 2]], scalars_are_mont=True))
        assert np.array_equal(batch[0], want)
        pp.release()
    finally:
        del os.environ["MZK_MSM_TABLE_BUDGET"]
    # ... and with the budget lifted the same SRS length gets its table and the same point
    pp = mj.UnivariateProverParam.gen_srs_for_testing(c, beta, n - 1)
    got = mj.jacobian_to_affine(c, mj.msm_bigint(pp, torch.from_numpy(x.view(np.int64)).cuda(), scalars_are_mont=True)[None])[0]
    mj.lib.check(L.mzk_srs_hbm_bytes(pp.handle, C.byref(pts), C.byref(tab)), "mzk_srs_hbm_bytes")
    mj.lib.check(L.mzk_srs_precompute(pp.handle, C.byref(bits), C.byref(levels), C.byref(nbytes), C.byref(ms)), "mzk_srs_precompute")
    assert bits.value == 16 and levels.value >= 16 and tab.value == levels.value * n * 112 and np.array_equal(got, want)
    pp.release()


@pytest.mark.parametrize("curve_id,log_n,count", [(0, 12, 5), (0, 15, 6), (1, 14, 8), (1, 16, 3), (0, 10, 2)])
def test_fused_batches_of_small_msms_against_the_oracle(gpu, mj, cref, curve_id, log_n, count):
    """Batches of small table-path MSMs are sorted and accumulated as ONE problem over the concatenation of their bucket sets (msm_pre.cuh,
    PreMulti): different lengths, different point ranges (base offsets), an empty and an all-zero member; every sum against the C oracle's
    Pippenger and against MZK_MSM_NO_FUSE-style single calls."""
    import torch
    c = mj.params.CURVES[curve_id]
    n = 1 << log_n
    pp = mj.UnivariateProverParam.gen_srs_for_testing(c, 0x1234567 + log_n, n + 2)
    srs = pp.powers_of_g()
    polys, offs = [], []
    for k in range(count):
        ln = [n + 3, n, n // 2 + 1, 1024, n + 2, 7 * n // 8, 1500, n][k % 8]
        ln = min(ln, n + 3)
        off = [0, 3, 1, 0, 1, n // 8, 0, 2][k % 8]
        off = min(off, n + 3 - ln)
        x = mj.params.random_fr_mont(c, ln, seed=100 + k)
        if k == 2:
            x[:] = 0                                                   # the zero polynomial: infinity
        polys.append(x)
        offs.append(off)
    d = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
    jac = mj.kzg.msm_bigint_batch(pp, d, scalars_are_mont=True, base_offsets=offs)
    for k in range(count):
        want = cref.jac_to_affine(curve_id, cref.msm(curve_id, srs[offs[k]:offs[k] + len(polys[k])], polys[k], scalars_are_mont=True, threads=4))[0]
        assert np.array_equal(cref.jac_to_affine(curve_id, jac[k])[0], want), (k, len(polys[k]), offs[k])
    pp.release()


def test_heavy_bucket_scratch_grows_on_demand_and_the_group_runs_again(gpu, mj, cref):
    """Round 5: the level-1 sums of heavy buckets are sized for an eighth of the worst case; a batch whose scalars put EVERY entry into heavy
    buckets (all-equal scalars) overflows it, the library enlarges the scratch from the counters it reads with the results and runs the
    group again (csrc/msm.hip MSM_RETRY).  Same points as the oracle, from a freshly released workspace (the share starts again there),
    and uniform scalars afterwards leave the scratch as it is."""
    import ctypes as C
    c = mj.params.CURVES[0]
    n = (1 << 17) + 5
    bases = cref.g1_arith_bases(0, 0xbeef, 0x2b, n)
    pp = mj.UnivariateProverParam.from_affine(0, bases)
    pats = _patterns(mj, c, n)
    L = mj.load()
    ws = C.c_uint64()
    mj.lib.check(L.mzk_workspace_release(), "mzk_workspace_release")
    names = ["dense", "dense", "half", "dense", "few"]
    jac = mj.msm_bigint_batch(pp, [pats[k] for k in names], scalars_are_mont=True)           # uniform: no heavy bucket, the optimistic size holds
    want = {k: cref.jac_to_affine(0, cref.msm(0, bases, pats[k], scalars_are_mont=True, threads=8))[0] for k in set(names) | {"all_equal", "plus_minus", "small_limbs"}}
    for i, k in enumerate(names):
        assert np.array_equal(cref.jac_to_affine(0, jac[i])[0], want[k]), k
    mj.lib.check(L.mzk_workspace_hbm_bytes(C.byref(ws)), "mzk_workspace_hbm_bytes")
    before = ws.value
    skew = ["all_equal", "plus_minus", "all_equal", "small_limbs", "all_equal"]
    for _ in range(2):                                                                        # the second call finds the scratch large enough
        jac = mj.msm_bigint_batch(pp, [pats[k] for k in skew], scalars_are_mont=True)
        for i, k in enumerate(skew):
            assert np.array_equal(cref.jac_to_affine(0, jac[i])[0], want[k]), ("skew", k)
    mj.lib.check(L.mzk_workspace_hbm_bytes(C.byref(ws)), "mzk_workspace_hbm_bytes")
    assert ws.value > before, "the skewed batch must have enlarged the heavy-bucket scratch"
    pp.release()


_LAUNCH_AB_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/oracle")
sys.path.insert(0, sys.argv[1] + "/tests")
import mpc_jellyfish_amd as mj
import cref
from importlib import import_module
import_module("mpc-jellyfish_amd.lib").init(0)
from test_msm_sparse_gpu import _patterns
c = mj.params.CURVES[0]
n = (1 << 16) + 3
bases = cref.g1_arith_bases(0, 0xbeef, 0x2b, n)
pp = mj.UnivariateProverParam.from_affine(0, bases)
pats = _patterns(mj, c, n)
L = mj.load()
for table in (1, 0):
    L.mzk_msm_set_precompute(table)
    for k in sorted(pats):
        print(table, k, hashlib.sha256(cref.jac_to_affine(0, mj.msm_bigint(pp, pats[k], scalars_are_mont=True)).tobytes()).hexdigest())
    names = ["dense", "all_equal", "half", "small_limbs", "plus_minus"]
    jac = mj.msm_bigint_batch(pp, [pats[k] for k in names], scalars_are_mont=True)
    for i, k in enumerate(names):
        print(table, "batch:" + k, hashlib.sha256(cref.jac_to_affine(0, jac[i]).tobytes()).hexdigest())
"""


def test_launch_diet_and_round_4_launch_sequence_give_the_same_points(gpu):
    """Round 5 folded six launches of an MSM into their neighbours (csrc/msm.hip `diet`); MZK_MSM_LEGACY_LAUNCHES=1 keeps the round-4
    sequence for A/B.  Both sequences -- each in a child process, the switch is read once -- on dense, sparse and skewed scalars, both MSM
    paths, single calls and a batch: the same affine points (which the tests above pin to the oracle for the default)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"MZK_MSM_LEGACY_LAUNCHES": "1"}, {"MZK_MSM_FOLD2": "1"}):
        r = subprocess.run([sys.executable, "-c", _LAUNCH_AB_SCRIPT, root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if len(l.split()) == 3])
    assert len(outs[0]) == 2 * (7 + 5)
    assert outs[0] == outs[1], "launch diet vs the round-4 launch sequence"
    assert outs[0] == outs[2], "two reduction levels per launch (off by default)"
