"""UnivariateKzgPCS -- mirror of primitives/src/pcs/univariate_kzg/{mod,srs}.rs for the commit path:

    UnivariateProverParam{powers_of_g}      srs.rs:36-40   -> an SRS resident in HBM (handle)
    gen_srs_for_testing                     srs.rs:118-153 -> UnivariateProverParam.gen_srs_for_testing
    trim                                    srs.rs:77-93   -> UnivariateProverParam.trim
    commit / batch_commit                   mod.rs:90-131  -> UnivariateKzgPCS.commit / batch_commit
    <G1 as VariableBaseMSM>::msm_bigint     mod.rs:109-111 -> msm_bigint

Polynomials are (len,4) uint64 Montgomery coefficient arrays, low order first (DensePolynomial);
commitments are affine points x||y (Montgomery limbs), (0,0) = infinity.  Error behaviour follows
the reference: commit raises PCSError (InvalidParameters) when the degree exceeds the key.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .params import CurveParams, curve as _curve, fq_to_mont, int_to_limbs


class PCSError(Exception):
    """primitives/src/pcs/errors.rs:16-33 (InvalidParameters is the only variant raised here)."""


class Commitment:
    """pcs/structs.rs:16-19: a G1 affine point; .xy is (2, fq_limbs) uint64 Montgomery."""

    def __init__(self, curve: CurveParams, xy: np.ndarray):
        self.curve = curve
        self.xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(2, curve.fq_limbs)

    def is_infinity(self) -> bool:
        return not self.xy.any()

    def __eq__(self, other):
        return isinstance(other, Commitment) and self.curve is other.curve and np.array_equal(self.xy, other.xy)

    def __repr__(self):
        return f"Commitment({self.curve.name}, x[0]=0x{int(self.xy[0, 0]):016x})"


class UnivariateProverParam:
    """powers_of_g held on the GPU.  `offset`/`length` give a trimmed view of one registration."""

    def __init__(self, curve: CurveParams, handle: int, length: int, offset: int = 0, owner: bool = True):
        self.curve, self.handle, self.length, self.offset, self._owner = curve, handle, length, offset, owner

    @classmethod
    def from_affine(cls, curve, powers_of_g: np.ndarray) -> "UnivariateProverParam":
        c = _curve(curve)
        a = np.ascontiguousarray(powers_of_g, dtype=np.uint64).reshape(-1, 2, c.fq_limbs)
        L = _lib.ensure_init()
        h = C.c_uint64()
        _lib.check(L.mzk_srs_register(c.curve_id, C.c_void_p(a.ctypes.data), a.shape[0], C.byref(h)), "mzk_srs_register")
        return cls(c, h.value, a.shape[0])

    @classmethod
    def gen_srs_for_testing(cls, curve, beta: int, max_degree: int, g=None) -> "UnivariateProverParam":
        """powers_of_g = [beta^i * g] for i <= max_degree (srs.rs:118-153), built on the GPU.  g: affine base point as canonical
        integers (x, y) -- `universal_setup_for_testing` draws it with G1::rand after beta (snark.rs:495-497); None: the curve's
        standard generator."""
        c = _curve(curve)
        L = _lib.ensure_init()
        h = C.c_uint64()
        b = int_to_limbs(beta % c.r, 4)
        gm = None if g is None else np.ascontiguousarray(np.concatenate([fq_to_mont(c, [g[0]])[0], fq_to_mont(c, [g[1]])[0]]))
        _lib.check(L.mzk_srs_generate_for_testing_g(c.curve_id, C.c_void_p(b.ctypes.data), None if gm is None else C.c_void_p(gm.ctypes.data),
                                                    max_degree + 1, C.byref(h)), "mzk_srs_generate_for_testing_g")
        return cls(c, h.value, max_degree + 1)

    @classmethod
    def gen_lagrange_srs_for_testing(cls, curve, beta: int, domain_size: int, n_extra: int = 3, g=None) -> "UnivariateProverParam":
        """The testing SRS over the Lagrange basis of the gate domain H (|H| = domain_size): point i = L_i(beta) g, then n_extra points
        beta^j (beta^n - 1) g (include/mzk.h, mzk_srs_generate_lagrange_for_testing).  An MSM of (values on H, blinders) over it equals
        the commitment of the masked coefficient form over gen_srs_for_testing(beta, ..., g) -- round 1 commits the wires this way
        (TurboPlonkProver.lagrange_ck): witness values are mostly small numbers, whose high digits cost the MSM nothing."""
        c = _curve(curve)
        assert domain_size & (domain_size - 1) == 0
        L = _lib.ensure_init()
        h = C.c_uint64()
        b = int_to_limbs(beta % c.r, 4)
        gm = None if g is None else np.ascontiguousarray(np.concatenate([fq_to_mont(c, [g[0]])[0], fq_to_mont(c, [g[1]])[0]]))
        _lib.check(L.mzk_srs_generate_lagrange_for_testing(c.curve_id, C.c_void_p(b.ctypes.data), None if gm is None else C.c_void_p(gm.ctypes.data),
                                                           domain_size.bit_length() - 1, n_extra, C.byref(h)), "mzk_srs_generate_lagrange_for_testing")
        return cls(c, h.value, domain_size + n_extra)

    def lagrange_key(self, domain_size: int, n_extra: int = 3) -> "UnivariateProverParam":
        """The Lagrange-basis key of THIS SRS for the gate domain of `domain_size` points, without the trapdoor: the inverse group-NTT of
        its first domain_size points (mzk_srs_lagrange_from_srs; 0.14 s at 2^16, 1.2 s at 2^20 on BLS12-381: a one-off per SRS and domain)."""
        assert domain_size & (domain_size - 1) == 0 and self.offset == 0 and self.length >= domain_size + n_extra
        h = C.c_uint64()
        _lib.check(_lib.ensure_init().mzk_srs_lagrange_from_srs(self.handle, domain_size.bit_length() - 1, n_extra, C.byref(h)), "mzk_srs_lagrange_from_srs")
        return UnivariateProverParam(self.curve, h.value, domain_size + n_extra)

    def slice(self, first: int, count: int) -> "UnivariateProverParam":
        """A NEW registration holding the points [first, first + count) of this one (mzk_srs_slice): what a rank of a multi-GPU prover keeps
        of the commit key -- its own fixed-base table, 1 / G of the size, built with the window that suits the slice."""
        assert 0 <= first and first + count <= self.length
        h = C.c_uint64()
        _lib.check(_lib.ensure_init().mzk_srs_slice(self.handle, self.offset + first, count, C.byref(h)), "mzk_srs_slice")
        return UnivariateProverParam(self.curve, h.value, count)

    def trim(self, supported_degree: int) -> "UnivariateProverParam":
        """srs.rs:77-93: keep powers_of_g[..=supported_degree]."""
        if supported_degree + 1 > self.length:
            raise PCSError("InvalidParameters: supported degree larger than the SRS")
        return UnivariateProverParam(self.curve, self.handle, supported_degree + 1, self.offset, owner=False)

    def powers_of_g(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.length - first if count is None else count
        out = np.empty((count, 2, self.curve.fq_limbs), dtype=np.uint64)
        _lib.check(_lib.ensure_init().mzk_srs_download(self.handle, self.offset + first, count, C.c_void_p(out.ctypes.data)),
                   "mzk_srs_download")
        return out

    def release(self):
        if self._owner and self.handle:
            _lib.check(_lib.load().mzk_srs_release(self.handle), "mzk_srs_release")
            self.handle = 0


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _degree_and_leading_zeros(coeffs: np.ndarray):
    """(degree, num_leading_zeros) of a coefficient array; degree of the zero polynomial is 0."""
    nz = np.flatnonzero(coeffs.any(axis=1))
    if nz.size == 0:
        return 0, coeffs.shape[0]
    return int(nz[-1]), int(nz[0])


def msm_bigint(pp: UnivariateProverParam, bigints, base_offset: int = 0, scalars_are_mont: bool = False) -> np.ndarray:
    """sum_i bigints[i] * powers_of_g[base_offset + i] as a Jacobian point (3, fq_limbs), Montgomery.
    Uses min(len) of the two sides like ark-ec.  bigints: (n,4) uint64 host array or int64 CUDA tensor."""
    L = _lib.ensure_init()
    out = np.empty((3, pp.curve.fq_limbs), dtype=np.uint64)
    avail = max(0, pp.length - base_offset)
    if _is_torch(bigints):
        import torch
        if bigints.dtype != torch.int64 or not bigints.is_cuda or not bigints.is_contiguous() or bigints.shape[-1] != 4:
            raise ValueError("expected a contiguous int64 CUDA tensor of shape (n, 4)")
        n = min(bigints.shape[0], avail)
        st = torch.cuda.current_stream(bigints.device).cuda_stream
        _lib.check(L.mzk_msm_dev(pp.handle, pp.offset + base_offset, bigints.data_ptr(), n, int(scalars_are_mont),
                                 C.c_void_p(out.ctypes.data), st), "mzk_msm_dev")
        return out
    s = np.ascontiguousarray(bigints, dtype=np.uint64).reshape(-1, 4)
    n = min(s.shape[0], avail)
    _lib.check(L.mzk_msm(pp.handle, pp.offset + base_offset, C.c_void_p(s.ctypes.data), n, int(scalars_are_mont),
                         C.c_void_p(out.ctypes.data)), "mzk_msm")
    return out


class UnivariateKzgPCS:
    @staticmethod
    def commit(prover_param: UnivariateProverParam, poly) -> Commitment:
        """mod.rs:90-116: degree guard, skip low-order zero coefficients, MSM, into_affine.
        poly: (len, 4) Montgomery coefficients, host array or CUDA tensor."""
        pp = prover_param
        if _is_torch(poly) and poly.is_cuda and 0 < poly.shape[0] <= pp.length:
            # device-resident coefficients that fit the key: zero coefficients contribute nothing to the MSM, so skipping the
            # leading ones (mod.rs:106) needs no host copy
            jac = msm_bigint(pp, poly.contiguous(), 0, scalars_are_mont=True)
            return Commitment(pp.curve, jacobian_to_affine(pp.curve, jac[None])[0])
        if _is_torch(poly):
            poly = poly.cpu().numpy().view(np.uint64)
        coeffs = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        degree, lead = _degree_and_leading_zeros(coeffs)
        if degree > pp.length:                                             # mod.rs:98
            raise PCSError(f"InvalidParameters: poly degree {degree} is larger than allowed {pp.length}")
        body = coeffs[lead:degree + 1] if lead <= degree else coeffs[:0]
        if lead + body.shape[0] > pp.length:
            # degree == powers_of_g.len(): passes the reference's guard, then ark-ec truncates to min(len)
            body = body[:pp.length - lead]
        if not body.size:
            lead = 0                                                       # the zero polynomial (ark-poly keeps no coefficients for it)
        L = _lib.ensure_init()
        out = np.empty((2, pp.curve.fq_limbs), dtype=np.uint64)
        _lib.check(L.mzk_msm_affine(pp.handle, pp.offset + lead, C.c_void_p(body.ctypes.data) if body.size else None,
                                    body.shape[0], 1, C.c_void_p(out.ctypes.data)), "mzk_msm_affine")
        return Commitment(pp.curve, out)

    @staticmethod
    def batch_commit(prover_param: UnivariateProverParam, polys) -> list[Commitment]:
        """mod.rs:119-131.  The reference maps `commit` over the polynomials on Rayon workers; here
        the MSMs run back to back on the GPU and share one bucket-reduction phase (mzk_msm_batch)."""
        pp = prover_param
        bodies, offs = [], []
        for poly in polys:
            coeffs = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
            degree, lead = _degree_and_leading_zeros(coeffs)
            if degree > pp.length:
                raise PCSError(f"InvalidParameters: poly degree {degree} is larger than allowed {pp.length}")
            body = coeffs[lead:degree + 1] if lead <= degree else coeffs[:0]
            if lead + body.shape[0] > pp.length:
                body = body[:pp.length - lead]
            bodies.append(np.ascontiguousarray(body))
            offs.append(pp.offset + lead)
        jac = msm_bigint_batch(pp, bodies, offs, scalars_are_mont=True, absolute_offsets=True)
        xy = jacobian_to_affine(pp.curve, jac)
        return [Commitment(pp.curve, xy[i]) for i in range(len(bodies))]


    @staticmethod
    def open(prover_param: UnivariateProverParam, polynomial, point: int):
        """mod.rs:135-161: witness polynomial p(X) / (X - point) (remainder dropped), its commitment, and p(point).
        polynomial: (len, 4) Montgomery coefficients, numpy or CUDA tensor.  Returns (Commitment proof, evaluation int)."""
        import torch
        from . import poly as _poly
        pp = prover_param
        t = polynomial if _is_torch(polynomial) else torch.from_numpy(np.ascontiguousarray(polynomial, dtype=np.uint64).reshape(-1, 4).view(np.int64)).cuda()
        t = t.contiguous()
        if t.shape[0] == 0:
            return Commitment(pp.curve, np.zeros((2, pp.curve.fq_limbs), dtype=np.uint64)), 0
        ev = _poly.evaluate(pp.curve, t, point)[0]
        if t.shape[0] == 1:                                                # constant polynomial: zero witness
            return Commitment(pp.curve, np.zeros((2, pp.curve.fq_limbs), dtype=np.uint64)), ev
        witness = _poly.div_by_linear(pp.curve, t, point)
        proof = UnivariateKzgPCS.commit(pp, witness)
        return proof, ev

    @staticmethod
    def batch_open(prover_param: UnivariateProverParam, polynomials, points):
        """mod.rs:163-190 ("a naive approach"): open every polynomial at its own point."""
        if len(polynomials) != len(points):
            raise PCSError("InvalidParameters: poly length %d is different from points length %d" % (len(polynomials), len(points)))
        out = [UnivariateKzgPCS.open(prover_param, p, z) for p, z in zip(polynomials, points)]
        return [o[0] for o in out], [o[1] for o in out]


def jacobian_to_affine(curve, xyz: np.ndarray) -> np.ndarray:
    """(n,3,fq_limbs) Jacobian -> (n,2,fq_limbs) affine on the host (`into_affine`, mod.rs:111)."""
    c = _curve(curve)
    a = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(-1, 3, c.fq_limbs)
    out = np.empty((a.shape[0], 2, c.fq_limbs), dtype=np.uint64)
    _lib.check(_lib.load().mzk_g1_jacobian_to_affine(c.curve_id, C.c_void_p(a.ctypes.data), a.shape[0], C.c_void_p(out.ctypes.data)),
               "mzk_g1_jacobian_to_affine")
    return out


def msm_bigint_batch(pp: UnivariateProverParam, scalar_sets, base_offsets=None, scalars_are_mont: bool = False,
                     absolute_offsets: bool = False) -> np.ndarray:
    """Several MSMs over one SRS in one call -> (k,3,fq_limbs) Jacobian.  scalar_sets: host (n_i,4) uint64
    arrays, or int64 CUDA tensors (all of one kind)."""
    L = _lib.ensure_init()
    k = len(scalar_sets)
    out = np.empty((k, 3, pp.curve.fq_limbs), dtype=np.uint64)
    if k == 0:
        return out
    offs = [0] * k if base_offsets is None else list(base_offsets)
    if not absolute_offsets:
        offs = [pp.offset + o for o in offs]
    limit = pp.offset + pp.length
    lens = (C.c_uint64 * k)()
    offa = (C.c_uint64 * k)(*offs)
    ptrs = (C.c_void_p * k)()
    if _is_torch(scalar_sets[0]):
        import torch
        for i, t in enumerate(scalar_sets):
            if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous() or t.shape[-1] != 4:
                raise ValueError("expected contiguous int64 CUDA tensors of shape (n, 4)")
            lens[i] = min(t.shape[0], max(0, limit - offs[i]))
            ptrs[i] = t.data_ptr()
        st = torch.cuda.current_stream(scalar_sets[0].device).cuda_stream
        _lib.check(L.mzk_msm_batch_dev(pp.handle, k, ptrs, lens, offa, int(scalars_are_mont), C.c_void_p(out.ctypes.data), st),
                   "mzk_msm_batch_dev")
        return out
    keep = []
    for i, s in enumerate(scalar_sets):
        a = np.ascontiguousarray(s, dtype=np.uint64).reshape(-1, 4)
        keep.append(a)
        lens[i] = min(a.shape[0], max(0, limit - offs[i]))
        ptrs[i] = a.ctypes.data if a.size else None
    _lib.check(L.mzk_msm_batch(pp.handle, k, ptrs, lens, offa, int(scalars_are_mont), C.c_void_p(out.ctypes.data)), "mzk_msm_batch")
    return out
