#!/usr/bin/env python3
"""tools/dropin_time.py -- the library's share of one TurboPlonk proof in SHIM-ONLY mode: what the two call-site swaps of
INTEGRATION.md section 2 issue when nothing else of the Rust prover changes -- host pointers in, host pointers out.

Per proof of n gates (m = 8n), in the reference's order:
    round 1   6 x ifft(n)  (5 wires + public input, constraint_system.rs:1172, 1257)      5 x commit (n + 2 scalars)
    round 2   1 x ifft(n)  (z, constraint_system.rs:1221)                                 1 x commit (n + 3)
    round 3   25 x coset fft(8n) of <= n + 3 coefficients (13 selectors, 5 sigmas, 5 wires, z, pi; prover.rs:552-567)
              [the quotient closure itself stays on the CPU in this mode, prover.rs:605-659: NOT timed here]
              1 x coset ifft(8n) (prover.rs:672)                                          5 x commit (n + 3)
    round 5   2 x commit (n + 2) (opening proofs, univariate_kzg/mod.rs:148-155)
Bytes over PCIe: 7 x 64n + 25 x (32(n + 3) + 256n) + 512n + 13 x 32(n + 3).

Modes:  "pageable"  ordinary host memory (numpy), one call per polynomial (mzk_ntt / mzk_msm) -- the naive shim;
        "pinned"    the shim allocates its evaluation buffers with mzk_host_alloc (page-locked): same calls, DMA without staging;
        "batch"     pinned buffers + mzk_ntt_batch / mzk_msm_batch, which pipeline upload k+1 | transform k | download k-1;
        "four_site" two MORE call sites swapped, still host pointers: compute_prod_permutation_polynomial -> mzk_plonk_perm_product
                    (constraint_system.rs:1197-1223) and compute_quotient_polynomial -> mzk_plonk_quotient (prover.rs:512-673) over a
                    proving key registered once -- the 25 coset FFTs and their 6.7 GB of evaluation vectors never exist on the host:
                    6 ifft(n) + product + quotient + 13 commits, 1.5 GB over PCIe, and the two CPU-heavy loops are gone too.

    python tools/dropin_time.py [--log-n 20] [--reps 3]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pcie_bytes(n):
    return 7 * 64 * n + 25 * (32 * (n + 3) + 256 * n) + 512 * n + 13 * 32 * (n + 3)


class HostBuf:
    """(rows, 4) uint64 host array, pageable (numpy) or page-locked (mzk_host_alloc)."""

    def __init__(self, L, rows, pinned):
        self.L, self.rows, self.pinned = L, rows, pinned
        if pinned:
            p = C.c_void_p()
            rc = L.mzk_host_alloc(rows * 32, C.byref(p))
            if rc != 0:
                raise RuntimeError("mzk_host_alloc failed: %s" % L.mzk_last_error().decode())
            self.ptr = p.value
            self.a = np.ctypeslib.as_array((C.c_uint64 * (rows * 4)).from_address(self.ptr)).reshape(rows, 4)
        else:
            self.a = np.zeros((rows, 4), dtype=np.uint64)
            self.ptr = self.a.ctypes.data

    def free(self):
        if self.pinned and self.ptr:
            self.L.mzk_host_free(C.c_void_p(self.ptr))
            self.ptr = 0


def four_site_bytes(n):
    return 6 * 64 * n + (5 * 32 * n + 32 * n) + (7 * 32 * (n + 3) + 256 * n) + 13 * 32 * (n + 3)


def measure(mj, L, curve, log_n, mode, reps=2, srs=None, pk=None):
    """Returns {"ms": library wall time of the calls listed above, ...}.  The data are random field elements: the timing does not
    depend on their values."""
    c = curve
    n = 1 << log_n
    m = 8 * n
    pinned = mode in ("pinned", "batch", "four_site")
    own = srs is None
    if own:
        srs = mj.UnivariateProverParam.gen_srs_for_testing(c, 0x1234567, n + 2)
    rnd = mj.params.random_fr_mont(c, n + 3, seed=77)
    coset = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    cos_p = coset.ctypes.data_as(C.c_void_p)
    small = [HostBuf(L, n + 3, pinned) for _ in range(7)]                 # wires, pi, z coefficient buffers
    # evaluation buffers: the reference holds all 25 coset-evaluation vectors of round 3 at once (prover.rs:552-567), and so does a
    # shim that batches them; the one-call-per-polynomial modes cycle through a few
    big = [HostBuf(L, m, pinned) for _ in range(25 if mode == "batch" else (1 if mode == "four_site" else 4))]
    own_pk = False
    if mode == "four_site":
        if pk is None:
            own_pk = True
            fixed = mj.params.random_fr_mont(c, 18 * n, seed=31).reshape(18, n, 4)
            pk = mj.plonk.ProvingKeyDevice.register(c, n, list(fixed[:13]), list(fixed[13:]), [1, 2, 3, 4, 5], classes=mj.plonk.quotient_classes_needed(5, n, top=False))
            del fixed
        wires = HostBuf(L, 5 * n, pinned)
        wires.a[:] = np.tile(rnd[:n], (5, 1))
        polys = HostBuf(L, 7 * (n + 3), pinned)
        polys.a[:] = np.tile(rnd, (7, 1))
        chal = mj.params.fr_to_mont(c, [0x1234567, 0x89abcde, 0xf012345])
        cp = lambda i: chal[i].ctypes.data_as(C.c_void_p)
    for b in small:
        b.a[:] = rnd
    out = np.zeros(18, dtype=np.uint64)
    outs = np.zeros((5, 18), dtype=np.uint64)
    chk = lambda rc, what: (_ for _ in ()).throw(RuntimeError("%s: %s" % (what, L.mzk_last_error().decode()))) if rc != 0 else None

    def ntt(buf, in_len, lg, inverse, cos):
        chk(L.mzk_ntt(c.curve_id, C.c_void_p(buf.ptr), in_len, lg, inverse, cos), "mzk_ntt")

    def msm(buf, length):
        chk(L.mzk_msm(srs.handle, 0, C.c_void_p(buf.ptr), length, 1, out.ctypes.data_as(C.c_void_p)), "mzk_msm")

    def ntt_batch(bufs, in_lens, lg, inverse, cos):
        k = len(bufs)
        ptrs = (C.c_void_p * k)(*[b.ptr for b in bufs])
        lens = (C.c_uint64 * k)(*in_lens)
        chk(L.mzk_ntt_batch(c.curve_id, k, ptrs, lens, lg, inverse, cos), "mzk_ntt_batch")

    def msm_batch(bufs, lengths):
        k = len(bufs)
        ptrs = (C.c_void_p * k)(*[b.ptr for b in bufs])
        lens = (C.c_uint64 * k)(*lengths)
        chk(L.mzk_msm_batch(srs.handle, k, ptrs, lens, None, 1, outs.ctypes.data_as(C.c_void_p)), "mzk_msm_batch")

    def one_proof():
        if mode == "four_site":
            for i in range(6):
                ntt(small[i], n, log_n, 1, None)
            for i in range(5):
                msm(small[i], n + 2)
            chk(L.mzk_plonk_perm_product(pk.handle, C.c_void_p(wires.ptr), cp(1), cp(2), C.c_void_p(small[6].ptr)), "mzk_plonk_perm_product")
            msm(small[6], n + 3)
            chk(L.mzk_plonk_quotient(pk.handle, C.c_void_p(polys.ptr), n + 3, cp(0), cp(1), cp(2), C.c_void_p(big[0].ptr)), "mzk_plonk_quotient")
            for i in range(5):
                msm(small[i], n + 3)
            for i in range(2):
                msm(small[i], n + 2)
            return
        if mode == "batch":
            ntt_batch(small[:6], [n] * 6, log_n, 1, None)
            msm_batch(small[:5], [n + 2] * 5)
            ntt_batch(small[6:7], [n], log_n, 1, None)
            msm_batch(small[6:7], [n + 3])
            # 25 forward coset NTTs: the shim copies each polynomial's coefficients into an evaluation buffer first
            for j in range(25):
                big[j].a[:n + 3] = small[j % 7].a
            ntt_batch(big, [n + 3] * 25, log_n + 3, 0, cos_p)
            ntt_batch(big[:1], [m], log_n + 3, 1, cos_p)
            msm_batch(small[:5], [n + 3] * 5)
            msm_batch(small[:2], [n + 2] * 2)
            return
        for i in range(6):
            ntt(small[i], n, log_n, 1, None)
        for i in range(5):
            msm(small[i], n + 2)
        ntt(small[6], n, log_n, 1, None)
        msm(small[6], n + 3)
        for i in range(25):
            b = big[i % len(big)]
            b.a[:n + 3] = small[i % 7].a
            ntt(b, n + 3, log_n + 3, 0, cos_p)
        ntt(big[0], m, log_n + 3, 1, cos_p)
        for i in range(5):
            msm(small[i], n + 3)
        for i in range(2):
            msm(small[i], n + 2)

    one_proof()                                                           # warm-up: plans, SRS table, staging buffers
    t0 = time.perf_counter()
    for _ in range(reps):
        one_proof()
    ms = (time.perf_counter() - t0) / reps * 1e3
    # the shim's own coefficient copies into the evaluation buffers (25 x 32(n+3) bytes of memcpy) are inside the figure: small
    for b in small + big:
        b.free()
    if mode == "four_site":
        wires.free()
        polys.free()
        if own_pk:
            pk.release()
    if own:
        srs.release()
    gb = (four_site_bytes(n) if mode == "four_site" else pcie_bytes(n)) / 1e9
    return {"mode": mode, "log_n": log_n, "ms": round(ms, 1), "pcie_gb": round(gb, 2), "pcie_gb_per_s": round(gb / (ms * 1e-3), 1),
            "calls": ("6 ifft(n) + mzk_plonk_perm_product + mzk_plonk_quotient + 13 msm, host pointers" if mode == "four_site" else
                      "7 ifft(n) + 25 coset fft(8n) + 1 coset ifft(8n) through host pointers, 13 msm with host scalars")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--modes", default="pageable,pinned,batch,four_site")
    args = ap.parse_args()
    import mpc_jellyfish_amd as mj
    from importlib import import_module
    L = import_module("mpc-jellyfish_amd.lib").init(0)
    res = [measure(mj, L, mj.params.BLS12_381, args.log_n, mode, args.reps) for mode in args.modes.split(",")]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
