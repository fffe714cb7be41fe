"""Multi-GPU sharding of the commit path; one process per GPU, torch.distributed.

  * independent MSMs (batch_commit, mod.rs:125-127): polynomial i goes to rank i % world -- no
    collective on the data path; the W affine results are gathered as plain bytes at the end.
  * one large MSM sharded by point range: rank g owns bases/scalars [g*N/G, (g+1)*N/G), computes a
    full Pippenger over its range and emits one Jacobian partial; the partials (<= 8 x 144 B) are
    all-gathered and summed locally on every rank (RCCL has no EC-add reduction op).
The collective is latency-bound (<= 1152 B); link bandwidth is irrelevant.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .params import curve as _curve


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of rank's share of n points (the first n % world ranks take one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def poly_owner(i: int, world: int) -> int:
    return i % world


def sum_jacobian(curve, points: np.ndarray) -> np.ndarray:
    """Host-side sum of Jacobian points (n, 3, fq_limbs) -> (3, fq_limbs); needs no GPU."""
    c = _curve(curve)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 3, c.fq_limbs)
    out = np.empty((3, c.fq_limbs), dtype=np.uint64)
    _lib.check(_lib.load().mzk_g1_sum_jacobian(c.curve_id, C.c_void_p(pts.ctypes.data), pts.shape[0],
                                                C.c_void_p(out.ctypes.data)), "mzk_g1_sum_jacobian")
    return out


def gather_partials(part: np.ndarray, group=None, device=None) -> np.ndarray:
    """The collective of one commit group: every rank's (k, 3, L) Jacobian partials -> (world, k, 3, L) on every rank.  ONE flat
    tensor, ONE all_gather_into_tensor (RCCL: the payload in a CUDA tensor; gloo: host tensors, no copies at all) and ONE transfer back.
    The partials are BORN on the host -- an MSM ends with its Horner tail on a CPU core (csrc/msm.hip) -- so a device-side sum of the
    gathered points would not save a hop: over RCCL it is H2D, collective, D2H either way (k x world x 144 B); the <= 8 additions per
    commitment run on the host in a microsecond each (mzk_g1_sum_jacobian).  tools/collective_time.py times this path."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = np.ascontiguousarray(part, dtype=np.uint64).view(np.int64).reshape(-1)
    t = torch.from_numpy(flat)
    if device is not None:
        t = t.to(device, non_blocking=True)
    out = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(out, t, group=group)
    except (RuntimeError, NotImplementedError):                            # a backend without the flat form: the list form
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        out = torch.cat(parts)
    return out.cpu().numpy().view(np.uint64).reshape((world,) + tuple(np.shape(part)))


def all_gather_sum(curve, partial_xyz: np.ndarray, group=None, device=None) -> np.ndarray:
    """All-gather every rank's Jacobian partial and add them up locally ("all-reduce" of EC sums).
    With backend nccl (= RCCL) the 144-byte payload travels in a CUDA tensor; gloo uses host tensors."""
    c = _curve(curve)
    part = np.ascontiguousarray(partial_xyz, dtype=np.uint64).reshape(1, 3, c.fq_limbs)
    return sum_jacobian(c, gather_partials(part, group, device)[:, 0])


class ShardedCommitter:
    """`UnivariateKzgPCS::{commit, batch_commit}` (mod.rs:90-131) across the ranks of a process group.

    Every polynomial is sharded by point range: rank g runs ONE fused batch of k MSMs over its slice
    [g*len/G, (g+1)*len/G) of every polynomial (base_offset = slice start into the replicated SRS and its
    precomputed table), then the k x G Jacobian partials (144 B each) are all-gathered and summed on every rank, so
    all ranks hold identical commitments and their transcripts stay in step.  Compared with "polynomial i on rank
    i % G"  the load is balanced for any k and G (5 wire commitments on 8 GPUs), at the price of
    the same one small collective.
    `msm_batch(ck, scalars_list, base_offsets) -> (k, 3, fq_limbs)` defaults to the device path; the CPU tests inject
    the oracle there."""

    def __init__(self, curve, ck, group=None, device=None, msm_batch=None, slice_srs: bool = False):
        """slice_srs: register this rank's point range of the SRS as an SRS of its own.  The fixed-base table's window is chosen
        by SRS size (csrc/msm.hip srs_build_pre_t): a 2^17-point slice gets a small window and 2^15 buckets per MSM instead of the
        full SRS's 2^19 -- at 8 GPUs the bucket reduction, not the accumulation, is what a shard's MSM costs -- and the rank
        holds 1 / G of the table."""
        self.c = _curve(curve)
        self.ck, self.group, self.device = ck, group, device
        self.msm_batch = msm_batch
        self.slice_srs = slice_srs and ck is not None and msm_batch is None
        self._slice = None

    def _local(self, slices, offsets):
        if self.msm_batch is not None:
            return self.msm_batch(self.ck, slices, offsets)
        from . import kzg
        if self._slice is not None:
            lo = self._slice_lo
            return kzg.msm_bigint_batch(self._slice, slices, [o - lo for o in offsets], scalars_are_mont=True)
        return kzg.msm_bigint_batch(self.ck, slices, offsets, scalars_are_mont=True)

    def commit_jacobian(self, polys) -> np.ndarray:
        """polys: list of (len, 4) Montgomery coefficient arrays / CUDA tensors, identical on every rank.
        Returns (k, 3, fq_limbs) Jacobian commitments, identical on every rank."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        k, L = len(polys), self.c.fq_limbs
        # ONE partition of the point indices for all polynomials -- by the SRS length when there is an SRS (so that a rank always
        # works on the same points and can keep just those), else by the longest polynomial of the call
        total = self.ck.length if self.ck is not None else max([int(p.shape[0]) for p in polys] + [1])
        lo_r, hi_r = shard_range(total, rank, world)
        if self.slice_srs and self._slice is None and hi_r > lo_r:
            self._slice = self.ck.slice(lo_r, hi_r - lo_r)               # mzk_srs_slice: a device copy of this rank's range
            self._slice_lo = lo_r
        slices, offsets = [], []
        for p in polys:
            lo, hi = min(lo_r, int(p.shape[0])), min(hi_r, int(p.shape[0]))
            s = p[lo:hi]
            slices.append(s.contiguous() if hasattr(s, "contiguous") else np.ascontiguousarray(s))
            offsets.append(lo if hi > lo else lo_r)
        part = np.ascontiguousarray(self._local(slices, offsets), dtype=np.uint64).reshape(k, 3, L)
        stacked = gather_partials(part, self.group, self.device)
        return np.stack([sum_jacobian(self.c, stacked[:, i]) for i in range(k)])

    # ---- coefficient-range mode: the pointwise stages of rounds 4 and 5 run on this rank's
    # ---- coefficient range only -- the very range its MSM shard needs -- with small exchanges of field elements
    def world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group)

    def rank(self) -> int:
        import torch.distributed as dist
        return dist.get_rank(self.group)

    def point_range(self):
        """[lo, hi) of the SRS indices this rank commits over (the fixed partition commit_jacobian uses)."""
        import torch.distributed as dist
        return shard_range(self.ck.length, dist.get_rank(self.group), dist.get_world_size(self.group))

    def all_gather_fr(self, values) -> list:
        """Every rank's list of field elements (canonical ints, same count on every rank) -> [rank][i]; 32 bytes per element."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        k = len(values)
        t = torch.tensor([(int(v) >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for v in values for j in range(4)], dtype=torch.uint64).view(torch.int64)
        if self.device is not None:
            t = t.to(self.device)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=self.group)
        out = []
        for p in parts:
            w = p.cpu().numpy().view(np.uint64).reshape(k, 4)
            out.append([sum(int(w[i, j]) << (64 * j) for j in range(4)) for i in range(k)])
        return out

    def commit_jacobian_slices(self, slices) -> np.ndarray:
        """Like commit_jacobian, for polynomials of which this rank holds ONLY its coefficient range: slices[i] = coefficients
        [lo, lo + len(slices[i])) of polynomial i, lo = point_range()[0] (an empty slice: nothing of it falls into the range)."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        k, L = len(slices), self.c.fq_limbs
        lo_r, hi_r = self.point_range()
        if self.slice_srs and self._slice is None and hi_r > lo_r:
            self._slice = self.ck.slice(lo_r, hi_r - lo_r)               # mzk_srs_slice: a device copy of this rank's range
            self._slice_lo = lo_r
        sl = [s.contiguous() if hasattr(s, "contiguous") else np.ascontiguousarray(s) for s in slices]
        assert all(int(s.shape[0]) <= hi_r - lo_r for s in sl)
        part = np.ascontiguousarray(self._local(sl, [lo_r] * k), dtype=np.uint64).reshape(k, 3, L)
        stacked = gather_partials(part, self.group, self.device)
        return np.stack([sum_jacobian(self.c, stacked[:, i]) for i in range(k)])

    def release(self):
        if self._slice is not None:
            self._slice.release()
            self._slice = None


def class_range(rank: int, world: int, n_classes: int = 8) -> list[int]:
    """Residue classes of the quotient domain owned by `rank` when the first `n_classes` classes are evaluated
    (plonk.quotient_classes_needed: 6 for TurboPlonk, 7 for UltraPlonk, 8 for tiny domains): contiguous blocks of
    ceil(n_classes / world), so that an all-gather in rank order is class-major; the last ranks may own fewer classes, or none."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank outside the world")
    per = -(-n_classes // world)
    return list(range(min(rank * per, n_classes), min((rank + 1) * per, n_classes)))


def gather_quotient_classes(local, group=None, via_host: bool = False, n_classes: int | None = None, per_rank: int | None = None):
    """The one exchange step of the chunked quotient: every rank contributes its class remainders and receives all of them,
    (n_classes, n, 4) class-major.  Ranks own ceil(n_classes / G) classes (class_range) except the last ones, which pad their
    contribution to that size so that one fixed-size all-gather serves (the padding is dropped on arrival).
    RCCL: all_gather_into_tensor on device tensors (n * 32 * per_rank bytes per rank over xGMI); gloo (CPU rehearsal): staged
    through host memory.  n_classes defaults to (classes per rank) x G -- the even split."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if n_classes is None:
        n_classes = local.shape[0] * world
    per = -(-n_classes // world) if per_rank is None else per_rank
    if local.shape[0] < per:                                              # a rank with fewer classes (or none) pads
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad]) if local.shape[0] else pad
    out_shape = (per * world,) + tuple(local.shape[1:])
    if via_host:
        h = local.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        return torch.cat(parts)[:n_classes].to(local.device)
    out = torch.empty(out_shape, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_classes]
