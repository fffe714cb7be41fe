"""Multi-GPU sharding; one process per GPU, torch.distributed.

  * `TorchComm`: the `mzk_comm` callbacks of the round-level C ABI over a process group -- a `prover.TurboPlonkProver` created with
    it is one rank of a sharded proof: every commitment by point range (rank g commits over its range of the SRS, a slice it may keep
    alone: kzg.UnivariateProverParam.slice), the quotient by residue class with one exchange, rounds 4-5 by coefficient range -- the
    decomposition lives in the library's rounds (csrc/prover.hip), this class only moves the bytes.
  * one large MSM sharded by point range (bench.py's weak-scaling headline): rank g computes a full Pippenger over its range and
    emits one Jacobian partial; the partials (<= 8 x 144 B) are all-gathered and summed locally on every rank (RCCL has no EC-add
    reduction op): `all_gather_sum`, `gather_partials`.
The collectives are latency-bound (<= 1152 B); link bandwidth matters for the one class exchange only.
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


class TorchComm:
    """`mzk_comm` (include/mzk.h) over torch.distributed: what makes a `prover.TurboPlonkProver` ONE RANK of a multi-process proof
    (one process per GPU).  The library's rounds call back for (i) the small all-gathers of Jacobian partials and
    partial evaluations (host buffers of a few hundred bytes), (ii) the one exchange of quotient-class remainders of round 3 (device
    buffers, n x 32 B per class) and (iii) barriers.  RCCL (backend "nccl"): payloads travel in CUDA tensors, the class exchange is
    one all_gather_into_tensor between staging tensors filled / drained by device-to-device copies; gloo (CPU rehearsal): host tensors,
    the classes staged through host memory.  Every rank must make the same calls in the same order -- the rounds do."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.group, self.device = group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        AG = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
        BA = C.CFUNCTYPE(C.c_int32, C.c_void_p)
        EX = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32)

        class Comm(C.Structure):
            _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32), ("all_gather", AG), ("barrier", BA), ("exchange_classes", EX)]

        self._cb = (AG(self._all_gather), BA(self._barrier), EX(self._exchange))            # (kept alive with the struct)
        self._struct = Comm(None, self.rank, self.world, *self._cb)

    def struct_ptr(self):
        return C.cast(C.pointer(self._struct), C.c_void_p)

    def _guard(self, fn, *args) -> int:
        try:
            fn(*args)
            return 0
        except Exception as e:                                         # noqa: BLE001  (a Python exception must not cross the C boundary)
            import sys
            print("TorchComm callback failed: %r" % (e,), file=sys.stderr)
            return 1

    def _all_gather(self, ctx, send, nbytes, recv):
        def run():
            import torch
            import torch.distributed as dist
            t = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone() if nbytes else torch.empty(0, dtype=torch.uint8)
            if self.device is not None:
                t = t.to(self.device)
            out = torch.empty(self.world * nbytes, dtype=torch.uint8, device=t.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
            host = out.cpu()                                             # (kept in a name until the memmove is done: with device payloads .cpu() makes a temporary)
            C.memmove(recv, host.data_ptr(), self.world * nbytes)
        return self._guard(run)

    def _barrier(self, ctx):
        import torch.distributed as dist
        return self._guard(lambda: dist.barrier(group=self.group))

    def _exchange(self, ctx, d_rem, class_bytes, first_own, n_own, n_classes):
        """the class remainders of all ranks resident in d_rem (n_classes x class_bytes on the device; this rank has filled
        [first_own, first_own + n_own)): ranks own contiguous blocks of ceil(n_classes / world) classes (class_range), the last may own fewer"""
        def run():
            import torch
            import torch.distributed as dist
            L = _lib.load()
            per = -(-n_classes // self.world)
            on_dev = self.device is not None
            local = torch.zeros(per * class_bytes, dtype=torch.uint8, device=self.device if on_dev else "cpu")
            mine = C.c_void_p(d_rem + first_own * class_bytes)
            if n_own:
                if on_dev:
                    _lib.check(L.mzk_dev_copy(C.c_void_p(local.data_ptr()), mine, n_own * class_bytes, None), "mzk_dev_copy")
                    _lib.check(L.mzk_dev_sync(), "mzk_dev_sync")
                else:
                    _lib.check(L.mzk_dev_download(C.c_void_p(local.data_ptr()), mine, n_own * class_bytes), "mzk_dev_download")
            out = torch.empty(self.world * per * class_bytes, dtype=torch.uint8, device=local.device)
            dist.all_gather_into_tensor(out, local, group=self.group)
            total = n_classes * class_bytes                              # rank order IS class order; the tail beyond n_classes is padding
            if on_dev:
                torch.cuda.synchronize()
                _lib.check(L.mzk_dev_copy(C.c_void_p(d_rem), C.c_void_p(out.data_ptr()), total, None), "mzk_dev_copy")
                _lib.check(L.mzk_dev_sync(), "mzk_dev_sync")
            else:
                _lib.check(L.mzk_dev_upload(C.c_void_p(d_rem), C.c_void_p(out.data_ptr()), total), "mzk_dev_upload")
        return self._guard(run)


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
