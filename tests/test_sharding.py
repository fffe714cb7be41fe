"""CPU suite: the N > 1 path.  Two gloo ranks shard one MSM by point range, all-gather their
Jacobian partials and add them up with the library's host-side EC sum (no GPU involved: the
per-rank partial MSM comes from the oracle here, exactly where the HIP MSM sits on a GPU box)."""
import os
import sys

import numpy as np
import pytest

import mirror_prover as MP          # ShardedCommitter: the torch.distributed committer of the test-side prover (tests/mirror_prover.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, curve_id, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import cref
    import mpc_jellyfish_amd as mj
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = mj.params.CURVES[curve_id]
        bases = cref.g1_arith_bases(curve_id, 4242, 17, n)                 # same on every rank (seeded)
        scalars = mj.params.random_fr_mont(c, n, seed=77)
        lo, hi = mj.sharding.shard_range(n, rank, world)
        partial = cref.msm(curve_id, bases[lo:hi], scalars[lo:hi]) if hi > lo else cref.msm(curve_id, bases[:0], scalars[:0])
        total = mj.sharding.all_gather_sum(c, partial)
        np.save(os.path.join(out_dir, f"total_{rank}.npy"), total)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("curve_id,n", [(0, 257), (1, 64), (0, 1)])
def test_point_range_sharded_msm_two_ranks(tmp_path, cref, mj, curve_id, n):
    import torch.multiprocessing as mp
    world = 2
    port = 29500 + (os.getpid() + n) % 2000
    mp.spawn(_worker, args=(world, port, curve_id, n, str(tmp_path)), nprocs=world, join=True)
    c = mj.params.CURVES[curve_id]
    bases = cref.g1_arith_bases(curve_id, 4242, 17, n)
    scalars = mj.params.random_fr_mont(c, n, seed=77)
    want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases, scalars))[0]
    for rank in range(world):
        total = np.load(tmp_path / f"total_{rank}.npy")
        assert np.array_equal(cref.jac_to_affine(curve_id, total)[0], want), rank


def test_shard_ranges_cover_exactly(mj):
    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 3, 4, 8):
            spans = [mj.sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [mj.sharding.poly_owner(i, 4) for i in range(6)] == [0, 1, 2, 3, 0, 1]


def test_host_ec_sum_edge_cases(mj, cref):
    """infinity operands, P + P (doubling branch), P + (-P)."""
    for curve_id in (0, 1):
        c = mj.params.CURVES[curve_id]
        L = c.fq_limbs
        P = cref.g1_mul_gen(curve_id, 5)
        one = cref.fq_convert(curve_id, np.array([[1] + [0] * (L - 1)], dtype=np.uint64), True)[0]
        jacP = np.stack([P[0], P[1], one])
        negP = cref.g1_mul_gen(curve_id, c.r - 5)
        jacN = np.stack([negP[0], negP[1], one])
        inf = np.stack([one, one, np.zeros(L, dtype=np.uint64)])
        s = mj.sharding.sum_jacobian(c, np.stack([inf, jacP, inf]))
        assert np.array_equal(cref.jac_to_affine(curve_id, s)[0], P)
        s = mj.sharding.sum_jacobian(c, np.stack([jacP, jacP]))
        assert np.array_equal(cref.jac_to_affine(curve_id, s)[0], cref.g1_mul_gen(curve_id, 10))
        s = mj.sharding.sum_jacobian(c, np.stack([jacP, jacN]))
        assert not s[2].any()
        s = mj.sharding.sum_jacobian(c, np.zeros((0, 3, L), dtype=np.uint64))
        assert not s[2].any()


def _committer_worker(rank, world, port, curve_id, lens, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import cref
    import mpc_jellyfish_amd as mj
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = mj.params.CURVES[curve_id]
        bases = cref.g1_arith_bases(curve_id, 99, 5, max(lens))
        polys = [mj.params.random_fr_mont(c, n, seed=300 + i) for i, n in enumerate(lens)]
        oracle_batch = lambda ck, slices, offs: np.stack([cref.msm(curve_id, bases[o:o + len(s)], s, scalars_are_mont=True) for s, o in zip(slices, offs)])
        com = MP.ShardedCommitter(c, None, msm_batch=oracle_batch)
        np.save(os.path.join(out_dir, f"commits_{rank}.npy"), com.commit_jacobian(polys))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batch_commit(tmp_path, cref, mj, world):
    """ShardedCommitter: k polynomials of different lengths (one empty, one shorter than the world size), every rank ends
    with the same k commitments = the unsharded MSMs."""
    import torch.multiprocessing as mp
    curve_id, lens = 0, [130, 131, 0, 1, 64]
    port = 29500 + (os.getpid() + 7 * world) % 2000
    mp.spawn(_committer_worker, args=(world, port, curve_id, lens, str(tmp_path)), nprocs=world, join=True)
    c = mj.params.CURVES[curve_id]
    bases = cref.g1_arith_bases(curve_id, 99, 5, max(lens))
    for i, n in enumerate(lens):
        want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[:n], mj.params.random_fr_mont(c, n, seed=300 + i), scalars_are_mont=True))[0]
        for rank in range(world):
            got = np.load(tmp_path / f"commits_{rank}.npy")[i]
            assert np.array_equal(cref.jac_to_affine(curve_id, got)[0], want), (i, rank)


@pytest.mark.parametrize("curve_id", [0, 1])
def test_host_batch_normalisation(mj, cref, curve_id):
    """mzk_g1_jacobian_to_affine (host only; `normalize_batch` / `.into_affine()`): one shared inversion over a batch that
    contains points at infinity at the ends and in the middle."""
    c = mj.params.CURVES[curve_id]
    pts = [cref.msm(curve_id, cref.g1_arith_bases(curve_id, 10 + i, 3, 4), mj.params.random_fr_mont(c, 4, seed=i)) for i in range(5)]
    inf = np.zeros_like(pts[0])
    batch = np.stack([inf, pts[0], pts[1], inf, inf, pts[2], pts[3], pts[4], inf])
    got = mj.jacobian_to_affine(c, batch)
    for i in range(batch.shape[0]):
        assert np.array_equal(got[i], cref.jac_to_affine(curve_id, batch[i])[0]), i
    assert mj.jacobian_to_affine(c, batch[:1]).any() == False and mj.jacobian_to_affine(c, batch[:0]).shape[0] == 0


def test_class_range_covers_the_needed_classes_for_every_world():
    """sharding.class_range: the 5 (TurboPlonk) / 6 (UltraPlonk) / 8 needed residue classes (6 / 7 without the top coefficients) over 1..8 ranks -- contiguous, disjoint,
    in rank order, every class owned exactly once; ranks beyond the classes own none (8 GPUs, 6 classes)."""
    from importlib import import_module
    sh = import_module("mpc-jellyfish_amd.sharding")
    pl = import_module("mpc-jellyfish_amd.plonk")
    # the device provers' rule: W classes, the W + 3 top coefficients come from the numerator (n > W + 2)
    assert pl.quotient_classes_needed(5, 1 << 20) == list(range(5)) and pl.quotient_classes_needed(6, 1 << 22) == list(range(6))
    assert pl.quotient_classes_needed(5, 8) == list(range(5)) and pl.quotient_classes_needed(5, 4) == list(range(8))
    assert pl.quotient_classes_needed(6, 8) == list(range(8)) and pl.quotient_classes_needed(6, 16) == list(range(6))
    # without them (host-pointer mzk_plonk_quotient, whose caller keeps the degree guard): W + 1 classes; n <= W + 3: all 8 (at n = W + 3
    # the expected degree is (W + 1) n - 1: no spare coefficient, the degree check could not fail)
    assert pl.quotient_classes_needed(5, 1 << 20, top=False) == list(range(6)) and pl.quotient_classes_needed(6, 1 << 22, top=False) == list(range(7))
    assert pl.quotient_classes_needed(5, 8, top=False) == list(range(8)) and pl.quotient_classes_needed(5, 4, top=False) == list(range(8))
    assert pl.quotient_classes_needed(5, 16, top=False) == list(range(6)) and pl.quotient_classes_needed(6, 16, top=False) == list(range(7))
    for ncl in (5, 6, 7, 8):
        for world in range(1, 9):
            owned = [sh.class_range(r, world, ncl) for r in range(world)]
            assert sum(owned, []) == list(range(ncl)), (ncl, world)
            per = -(-ncl // world)
            assert all(len(o) <= per for o in owned) and all(o == list(range(o[0], o[0] + len(o))) for o in owned if o)
    assert sh.class_range(7, 8, 6) == [] and sh.class_range(3, 4, 6) == [] and sh.class_range(3, 4, 7) == [6]
    with pytest.raises(ValueError):
        sh.class_range(2, 2)


def _gather_worker(rank, world, port, n_classes, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from importlib import import_module
    sh = import_module("mpc-jellyfish_amd.sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        own = sh.class_range(rank, world, n_classes)
        local = torch.stack([torch.full((5, 4), 100 * k + 1, dtype=torch.int64) for k in own]) if own else torch.empty((0, 5, 4), dtype=torch.int64)
        every = sh.gather_quotient_classes(local, via_host=True, n_classes=n_classes)
        torch.save(every, os.path.join(out_dir, f"classes_{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_classes", [(4, 6), (8, 6), (8, 7), (2, 7)])
def test_gather_of_unevenly_owned_classes(tmp_path, world, n_classes):
    """The one exchange of the chunked quotient when the classes do not divide evenly (4 ranks x 6 classes: 2, 2, 2, 0;
    8 ranks x 6 classes: two ranks own nothing): every rank receives the n_classes remainders, class-major, padding dropped."""
    import torch
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + 13 * world + n_classes) % 2000
    mp.spawn(_gather_worker, args=(world, port, n_classes, str(tmp_path)), nprocs=world, join=True)
    want = torch.stack([torch.full((5, 4), 100 * k + 1, dtype=torch.int64) for k in range(n_classes)])
    for rank in range(world):
        assert torch.equal(torch.load(tmp_path / f"classes_{rank}.pt"), want), rank


class _Srs:
    """what ShardedCommitter needs of a commit key on the CPU: its length (the fixed partition of the point indices)"""

    def __init__(self, length):
        self.length = length


def _range_worker(rank, world, port, curve_id, srs_len, poly_len, out_dir):
    """Rounds 4-5 by coefficient range (prover.py _RangeEvals / _openings_ranged) with Python integers in place of the device
    kernels: partial evaluations, the one carried coefficient per opening, the division on the extended range, the range commit."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import cref
    import mpc_jellyfish_amd as mj
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = mj.params.CURVES[curve_id]
        r = c.r
        bases = cref.g1_arith_bases(curve_id, 31, 7, srs_len)
        oracle_batch = lambda ck, slices, offs: np.stack([cref.msm(curve_id, bases[o:o + len(s)], np.asarray(s), scalars_are_mont=True) for s, o in zip(slices, offs)])
        com = MP.ShardedCommitter(c, _Srs(srs_len), msm_batch=oracle_batch)
        assert com.rank() == rank and com.world() == world
        b = mj.params.fr_from_mont(c, mj.params.random_fr_mont(c, poly_len, seed=5))       # the batch polynomial, same on every rank
        z = 0x1234567890abcdef % r
        lo, hi = com.point_range()
        hi = min(hi, poly_len)
        width = max(hi - lo, 0)
        own = b[lo:hi] if width else []
        e = sum(v * pow(z, j, r) for j, v in enumerate(own)) % r                       # the range read as a polynomial, at z
        every = com.all_gather_fr([e, e * pow(z, min(lo, poly_len), r) % r])
        value = sum(col[1] for col in every) % r                                      # round 4: p(z) from the partial values
        carry = 0
        for q in range(rank + 1, world):
            lo_q = min(mj.sharding.shard_range(srs_len, q, world)[0], poly_len)
            carry = (carry + pow(z, lo_q - hi, r) * every[q][0]) % r
        ext = own + [carry]
        wit, acc = [0] * width, 0
        for j in range(width, 0, -1):                                                  # synthetic division of the extended range by X - z
            acc = (ext[j] + z * acc) % r
            wit[j - 1] = acc
        jac = com.commit_jacobian_slices([mj.params.fr_to_mont(c, wit)])
        np.save(os.path.join(out_dir, f"open_{rank}.npy"), jac)
        np.save(os.path.join(out_dir, f"value_{rank}.npy"), mj.params.fr_to_mont(c, [value]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,srs_len,poly_len", [(2, 23, 19), (3, 23, 23), (3, 40, 9)])
def test_openings_by_coefficient_range(tmp_path, cref, mj, world, srs_len, poly_len):
    """Every rank divides only its coefficient range (plus one carried coefficient) and commits only that range; the summed
    commitment is the commitment of the whole witness polynomial b(X) / (X - z), and the partial evaluations add up to b(z).
    (3, 40, 9): the upper ranks' ranges lie beyond the polynomial."""
    import torch.multiprocessing as mp
    curve_id = 1
    port = 29500 + (os.getpid() + 17 * world + poly_len) % 2000
    mp.spawn(_range_worker, args=(world, port, curve_id, srs_len, poly_len, str(tmp_path)), nprocs=world, join=True)
    c = mj.params.CURVES[curve_id]
    r = c.r
    b = mj.params.fr_from_mont(c, mj.params.random_fr_mont(c, poly_len, seed=5))
    z = 0x1234567890abcdef % r
    wit, acc = [0] * (poly_len - 1), 0
    for j in range(poly_len - 1, 0, -1):
        acc = (b[j] + z * acc) % r
        wit[j - 1] = acc
    bases = cref.g1_arith_bases(curve_id, 31, 7, srs_len)
    want = cref.jac_to_affine(curve_id, cref.msm(curve_id, bases[:poly_len - 1], mj.params.fr_to_mont(c, wit), scalars_are_mont=True))[0]
    want_value = sum(v * pow(z, j, r) for j, v in enumerate(b)) % r
    for rank in range(world):
        got = np.load(tmp_path / f"open_{rank}.npy")[0]
        assert np.array_equal(cref.jac_to_affine(curve_id, got)[0], want), rank
        assert mj.params.fr_from_mont(c, np.load(tmp_path / f"value_{rank}.npy")) == [want_value], rank
