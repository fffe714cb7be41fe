"""Device-resident arithmetic of one TurboPlonk proof -- the five rounds of
`PlonkKzgSnark::batch_prove_internal` (plonk/src/proof_system/snark.rs:201-469) for a single instance
without Plookup, with every NTT, MSM, pointwise pass, scan and polynomial operation on the GPU and all
polynomials kept in HBM between rounds (SURVEY.md 8(f) N1 + N2).

    round 1  run_1st_round   prover.rs:72-87     wire iNTTs, masking, batch_commit, public-input iNTT
    round 2  run_2nd_round   prover.rs:125-141   permutation grand product, masking, commit
    round 3  run_3rd_round   prover.rs:192-209   quotient (coset NTTs + fused kernel + coset iNTT), split, batch_commit
    round 4  compute_evaluations                 prover.rs:216-235
    round 5  linearisation + opening proofs      prover.rs:302-358, 362-419, 490-509, 963-1035

Challenges come either from the caller (`ProverChallenges`: the way the reference's own per-round tests
fix them, multiprover/proof_system/prover.rs:1316-1556) or from the Merlin transcript mirror
(`transcript.StandardTranscript`, message order of snark.rs:263-431).  The blinding RNG (`test_rng`, ChaCha)
is not restated: blinding scalars are inputs.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import kzg, plonk, poly
from . import transcript as _transcript
from .domain import Radix2EvaluationDomain
from .params import CurveParams, curve as _curve, fr_to_mont


@dataclass
class Blinders:
    """DensePolynomial::rand draws, in the reference's order: 5 x 2 for the wires,
    3 for z, 4 for the split quotient."""
    wires: list
    z: list
    quot: list


@dataclass
class ProverChallenges:
    beta: int
    gamma: int
    alpha: int
    zeta: int
    v: int


class FixedChallenges:
    """Challenge source with externally fixed values."""

    def __init__(self, ch: ProverChallenges):
        self.ch = ch

    def after_round1(self, wires_comms):
        return self.ch.beta, self.ch.gamma

    def after_round2(self, z_comm):
        return self.ch.alpha

    def after_round3(self, split_comms):
        return self.ch.zeta

    def after_round4(self, wires_evals, wire_sigma_evals, perm_next_eval):
        return self.ch.v


class TranscriptChallenges:
    """Challenge source following batch_prove_internal (snark.rs:263-431) on a StandardTranscript."""

    def __init__(self, prover: "TurboPlonkProver", pub_input, extra_msg: bytes | None = None):
        c = prover.curve
        self.c = c
        self.t = _transcript.StandardTranscript(c, b"PlonkProof")
        if extra_msg is not None:
            self.t.append_message(b"extra info", extra_msg)
        sel, sig = prover.vk_commitments()
        self.t.append_vk_and_pub_input(prover.n, len(pub_input), prover.k, [self._pt(x) for x in sel], [self._pt(x) for x in sig], pub_input)
        self.challenges = {}

    def _pt(self, comm: kzg.Commitment):
        if comm.is_infinity():
            return None
        from .params import fq_from_mont
        x, y = fq_from_mont(self.c, comm.xy)
        return (x, y)

    def _squeeze(self, name: str) -> int:
        v = self.t.get_and_append_challenge(name.encode())
        self.challenges[name] = v
        return v

    def after_round1(self, wires_comms):
        self.t.append_commitments(b"witness_poly_comms", [self._pt(cm) for cm in wires_comms])
        self._squeeze("tau")                                   # squeezed even without Plookup (snark.rs:293)
        return self._squeeze("beta"), self._squeeze("gamma")

    def after_round2(self, z_comm):
        self.t.append_commitment(b"perm_poly_comms", self._pt(z_comm))
        return self._squeeze("alpha")

    def after_round3(self, split_comms):
        self.t.append_commitments(b"quot_poly_comms", [self._pt(cm) for cm in split_comms])
        return self._squeeze("zeta")

    def after_round4(self, wires_evals, wire_sigma_evals, perm_next_eval):
        for e in wires_evals:
            self.t.append_field_elem(b"wire_evals", e)
        for e in wire_sigma_evals:
            self.t.append_field_elem(b"wire_sigma_evals", e)
        self.t.append_field_elem(b"perm_next_eval", perm_next_eval)
        return self._squeeze("v")


@dataclass
class ProofCore:
    """Proof (structs.rs:62-84) minus serialisation: commitments as affine x||y limbs, evaluations as ints."""
    wires_poly_comms: list
    prod_perm_poly_comm: kzg.Commitment
    split_quot_poly_comms: list
    opening_proof: kzg.Commitment
    shifted_opening_proof: kzg.Commitment
    wires_evals: list
    wire_sigma_evals: list
    perm_next_eval: int
    timings_ms: dict = field(default_factory=dict)


class TurboPlonkProver:
    """Holds a proving key on the device: coefficient forms (for rounds 4-5), the resident coset
    evaluations (round 3) and the commit key."""

    def __init__(self, curve, domain_size: int, selector_polys, sigma_polys, k, commit_key: kzg.UnivariateProverParam):
        import torch
        self.curve: CurveParams = _curve(curve)
        self.n = domain_size
        self.log_n = domain_size.bit_length() - 1
        self.k = list(k)
        self.ck = commit_key
        self.pk = plonk.ProvingKeyDevice.register(self.curve, domain_size, selector_polys, sigma_polys, k)
        pad = lambda p: np.concatenate([np.asarray(p, dtype=np.uint64).reshape(-1, 4),
                                        np.zeros((domain_size - np.asarray(p).reshape(-1, 4).shape[0], 4), dtype=np.uint64)])
        self.fixed = torch.from_numpy(np.stack([pad(p) for p in list(selector_polys) + list(sigma_polys)]).view(np.int64)).cuda()
        self.domain = Radix2EvaluationDomain(self.curve, self.log_n)
        self.w_n = pow(self.curve.fr_generator, (self.curve.r - 1) >> self.log_n, self.curve.r)

    def vk_commitments(self):
        """selector_comms, sigma_comms of the verifying key (preprocess, snark.rs:562-594): 18 commits, cached."""
        if getattr(self, "_vk", None) is None:
            jac = kzg.msm_bigint_batch(self.ck, [self.fixed[i] for i in range(18)], scalars_are_mont=True)
            xy = kzg.jacobian_to_affine(self.curve, jac)
            self._vk = ([kzg.Commitment(self.curve, xy[i]) for i in range(13)], [kzg.Commitment(self.curve, xy[13 + i]) for i in range(5)])
        return self._vk

    def release(self):
        self.pk.release()

    def _mask(self, t, row, blinders):
        """poly + (b_0 + b_1 X + ..)(X^n - 1) on the device row (prover.rs:463-486)."""
        import torch
        b = torch.from_numpy(fr_to_mont(self.curve, blinders).view(np.int64)).to(t.device)
        nb = torch.from_numpy(fr_to_mont(self.curve, [(-x) % self.curve.r for x in blinders]).view(np.int64)).to(t.device)
        # coefficients 0..h of an iNTT output are arbitrary: add -b there needs a field addition -> lincomb on a slice
        h = len(blinders)
        head = t[row, :h].clone()
        one = 1
        poly.lincomb(self.curve, [(one, head), (one, nb)], out=t[row, :h])
        t[row, self.n:self.n + h] = b

    def prove(self, wire_values, pub_input_values, ch, blind: Blinders, profile: bool = False) -> ProofCore:
        """ch: ProverChallenges (fixed) or a challenge source (FixedChallenges / TranscriptChallenges)."""
        src = FixedChallenges(ch) if isinstance(ch, ProverChallenges) else ch
        import time
        import torch
        c, n, r = self.curve, self.n, self.curve.r
        m = 8 * n
        tm = {}

        def tick(name, t0):
            if profile:
                torch.cuda.synchronize()
                tm[name] = round((time.perf_counter() - t0) * 1e3, 3)

        dev = self.fixed.device
        wv = wire_values if hasattr(wire_values, "is_cuda") else torch.from_numpy(np.ascontiguousarray(wire_values).view(np.int64)).to(dev)
        pv = pub_input_values if hasattr(pub_input_values, "is_cuda") else torch.from_numpy(np.ascontiguousarray(pub_input_values).view(np.int64)).to(dev)
        # one slab for round 3: rows 0-4 wires, 5 z, 6 public input; coefficients in the first n+3 columns
        t0 = time.perf_counter()
        slab = torch.zeros((7, m, 4), dtype=torch.int64, device=dev)
        # ---- round 1 (prover.rs:72-87)
        coeff = torch.empty((6, n, 4), dtype=torch.int64, device=dev)
        coeff[:5] = wv
        coeff[5] = pv
        self.domain.ifft_in_place(coeff)
        slab[:5, :n] = coeff[:5]
        slab[6, :n] = coeff[5]
        for i in range(5):
            self._mask(slab, i, blind.wires[i])
        wire_polys = [slab[i, :n + 2] for i in range(5)]
        tick("r1_ntt_mask", t0)
        t0 = time.perf_counter()
        jac = kzg.msm_bigint_batch(self.ck, [p.contiguous() for p in wire_polys], scalars_are_mont=True)
        wires_comms = [kzg.Commitment(c, xy) for xy in kzg.jacobian_to_affine(c, jac)]
        tick("r1_commit", t0)
        # ---- round 2 (prover.rs:125-141; constraint_system.rs:1197-1223)
        t0 = time.perf_counter()
        beta, gamma = src.after_round1(wires_comms)
        bg = fr_to_mont(c, [beta, gamma])
        from . import lib as _lib
        import ctypes as C
        _lib.check(_lib.ensure_init().mzk_plonk_perm_product_dev(self.pk.handle, wv.data_ptr(), bg[0].ctypes.data_as(C.c_void_p),
                                                                 bg[1].ctypes.data_as(C.c_void_p), coeff[0].data_ptr(),
                                                                 torch.cuda.current_stream(dev).cuda_stream), "mzk_plonk_perm_product_dev")
        slab[5, :n] = coeff[0]
        self._mask(slab, 5, blind.z)
        z_poly = slab[5, :n + 3]
        tick("r2_product", t0)
        t0 = time.perf_counter()
        z_comm = kzg.Commitment(c, kzg.jacobian_to_affine(c, kzg.msm_bigint(self.ck, z_poly.contiguous(), scalars_are_mont=True))[0])
        tick("r2_commit", t0)
        # ---- round 3 (prover.rs:192-209, 512-673, 902-960)
        t0 = time.perf_counter()
        keep = slab[:6, :n + 3].clone()                                  # coefficient forms survive the in-place coset NTT
        quot = torch.empty((m, 4), dtype=torch.int64, device=dev)
        alpha = src.after_round2(z_comm)
        plonk.compute_quotient_polynomial_dev(self.pk, plonk.Challenges(alpha, beta, gamma), slab, n + 3, quot)
        tick("r3_quotient", t0)
        t0 = time.perf_counter()
        expected = 5 * (n + 1) + 2
        split = []
        last = 0
        for i in range(5):
            lo = i * (n + 2)
            hi = (i + 1) * (n + 2) if i < 4 else expected + 1
            p = torch.zeros((n + 3, 4), dtype=torch.int64, device=dev)
            p[:hi - lo] = quot[lo:hi]
            if i < 4:
                p[n + 2] = torch.from_numpy(fr_to_mont(c, [blind.quot[i]]).view(np.int64)).to(dev)[0]
            if last:
                negl = torch.from_numpy(fr_to_mont(c, [(-last) % r]).view(np.int64)).to(dev)
                poly.lincomb(c, [(1, p[:1].clone()), (1, negl)], out=p[:1])
            last = blind.quot[i] if i < 4 else 0
            split.append(p if i < 4 else p[:n])
        tick("r3_split", t0)
        t0 = time.perf_counter()
        jac = kzg.msm_bigint_batch(self.ck, [p.contiguous() for p in split], scalars_are_mont=True)
        split_comms = [kzg.Commitment(c, xy) for xy in kzg.jacobian_to_affine(c, jac)]
        tick("r3_commit", t0)
        # ---- round 4 (prover.rs:216-235)
        t0 = time.perf_counter()
        wire_polys = [keep[i, :n + 2] for i in range(5)]
        z_poly = keep[5]
        zeta = src.after_round3(split_comms)
        wires_evals = poly.evaluate(c, keep[:5], zeta, length=n + 2)
        wire_sigma_evals = poly.evaluate(c, self.fixed[13:17], zeta)
        perm_next_eval = poly.evaluate(c, z_poly, zeta * self.w_n % r)[0]
        tick("r4_evals", t0)
        # ---- round 5: linearisation polynomial (prover.rs:963-1035, 343-358) and openings (362-419, 490-509)
        t0 = time.perf_counter()
        v_ch = src.after_round4(wires_evals, wire_sigma_evals, perm_next_eval)
        we = wires_evals
        sel = self.fixed
        terms = [(we[j], sel[j]) for j in range(4)]
        terms += [(we[0] * we[1] % r, sel[4]), (we[2] * we[3] % r, sel[5])]
        terms += [(pow(we[j], 5, r), sel[6 + j]) for j in range(4)]
        terms += [(we[0] * we[1] % r * we[2] % r * we[3] % r * we[4] % r, sel[12]), ((-we[4]) % r, sel[10]), (1, sel[11])]
        vanish = (pow(zeta, n, r) - 1) % r
        lagrange_1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r
        cf = alpha
        for j in range(5):
            cf = cf * (we[j] + beta * self.k[j] % r * zeta + gamma) % r
        terms.append(((cf + alpha * alpha % r * lagrange_1) % r, z_poly))
        cf = alpha * beta % r * perm_next_eval % r
        for j in range(4):
            cf = cf * (we[j] + beta * wire_sigma_evals[j] + gamma) % r
        terms.append(((-cf) % r, self.fixed[17]))
        zeta_n2 = (vanish + 1) * zeta % r * zeta % r
        cf = 1
        for i in range(5):
            terms.append(((-vanish) * cf % r, split[i]))
            cf = cf * zeta_n2 % r
        lin = poly.lincomb(c, terms, out_len=n + 3)
        bterms = [(1, lin)]
        cf = v_ch
        for p in wire_polys + [self.fixed[13 + j] for j in range(4)]:
            bterms.append((cf, p))
            cf = cf * v_ch % r
        batch = poly.lincomb(c, bterms, out_len=n + 3)
        opening = poly.div_by_linear(c, batch, zeta)
        shifted = poly.div_by_linear(c, z_poly.contiguous(), zeta * self.w_n % r)
        tick("r5_polys", t0)
        t0 = time.perf_counter()
        jac = kzg.msm_bigint_batch(self.ck, [opening, shifted], scalars_are_mont=True)
        xy = kzg.jacobian_to_affine(c, jac)
        tick("r5_commit", t0)
        self.last = {"wire_polys": wire_polys, "z_poly": z_poly, "quot": quot, "split": split, "lin": lin, "opening": opening, "shifted": shifted}
        return ProofCore(wires_comms, z_comm, split_comms, kzg.Commitment(c, xy[0]), kzg.Commitment(c, xy[1]),
                         wires_evals, wire_sigma_evals, perm_next_eval, tm)
