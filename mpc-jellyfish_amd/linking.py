"""Proof linking on device-resident wire polynomials -- host-side mirror of the reference's prover half of the
proof-linking sub-protocol ("proof-linking commits through the same ABI"):

    GroupLayout                               relation/src/proof_linking/mod.rs:16-54
    LinkingHint                               plonk/src/proof_system/structs.rs:88-97   (made by snark.prove_with_link_hint)
    PlonkKzgSnark::link_proofs                plonk/src/proof_system/proof_linking.rs:80-111
    LinkingProof::serialize_compressed        proof_linking.rs:33-39

The reference expands Z_D(X) = prod (X - g^(offset+i)) and runs one dense long division (proof_linking.rs:119-158); here the
difference a_1 - a_2 stays on the device and mzk_poly_div_roots_dev divides it: when it vanishes on the link domain (a valid
link) by two coset NTTs around a pointwise 1 / Z_D(x), otherwise by the `size` linear factors one after another with the
synthetic division of round 5 -- floor division by a product equals the composition of the floor divisions by its
factors, so the quotient (remainder dropped, as ark-poly drops it) has the reference's coefficients in both cases.  The two
commitments are MSMs over the registered SRS (mzk_msm_affine).  There is no CPU path: without the HIP library every call
raises.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import kzg, poly
from . import transcript as _transcript
from .params import curve as _curve

PROOF_LINK_WIRE_IDX = 0                                     # relation/src/proof_linking/linkable_circuit.rs:23


@dataclass(frozen=True)
class GroupLayout:
    """The group is allocated on the 2^alignment-th roots of unity, `size` of them starting at `offset`."""
    alignment: int
    offset: int
    size: int

    def range_in_nth_roots(self, n: int):
        """inclusive (start, end) row range when embedded in the 2^n-th roots of unity (mod.rs:37-47)"""
        if n < self.alignment:
            raise ValueError("Group alignment must be <= n")
        spacing = 1 << (n - self.alignment)
        start = self.offset * spacing
        return start, start + max(self.size - 1, 0) * spacing

    def get_domain_generator(self, curve) -> int:
        c = _curve(curve)
        if self.alignment > c.two_adicity:
            raise ValueError("field 2-adicity too small for layout %r" % (self,))
        return pow(c.fr_generator, (c.r - 1) >> self.alignment, c.r)


@dataclass
class LinkingHint:
    linking_wire_poly: object                               # (len, 4) int64 CUDA tensor, Montgomery coefficients of the masked a(X)
    linking_wire_comm: kzg.Commitment


@dataclass
class LinkingProof:
    quotient_commitment: kzg.Commitment
    opening_proof: kzg.Commitment

    def serialize_compressed(self) -> bytes:
        from .snark import _g1
        c = self.quotient_commitment.curve
        return _g1(c, self.quotient_commitment) + _g1(c, self.opening_proof)


def _point(c, cm: kzg.Commitment):
    if cm.is_infinity():
        return None
    from .params import fq_from_mont
    x, y = fq_from_mont(c, cm.xy)
    return (x, y)


def compute_vanishing_poly_eval(curve, challenge: int, layout: GroupLayout) -> int:
    """proof_linking.rs:162-176"""
    c = _curve(curve)
    g = layout.get_domain_generator(c)
    root = pow(g, layout.offset, c.r)
    out = 1
    for _ in range(layout.size):
        out = out * ((challenge - root) % c.r) % c.r
        root = root * g % c.r
    return out


def compute_quotient_challenge(curve, a1_comm: kzg.Commitment, a2_comm: kzg.Commitment, quotient_comm: kzg.Commitment) -> int:
    """proof_linking.rs:185-197: eta from a transcript of its own that absorbs the two linking-wire commitments."""
    c = _curve(curve)
    t = _transcript.StandardTranscript(c, b"PlonkLinkingProof")
    t.append_commitments(b"linking_wire_comms", [_point(c, a1_comm), _point(c, a2_comm)])
    t.append_commitment(b"quotient_comm", _point(c, quotient_comm))
    return t.get_and_append_challenge(b"eta")


def compute_linking_quotient(curve, a1, a2, layout: GroupLayout):
    """(a_1 - a_2) / Z_D on the device (proof_linking.rs:119-134).  Returns (difference, quotient) as CUDA tensors."""
    c = _curve(curve)
    layout.get_domain_generator(c)                          # two-adicity check
    diff = poly.lincomb(c, [(1, a1), (c.r - 1, a2)])
    return diff, poly.div_by_roots_of_unity(c, diff, layout.alignment, layout.offset, layout.size)


def _commit_dev(ck, t) -> kzg.Commitment:
    if t.shape[0] == 0:
        return kzg.Commitment(ck.curve, np.zeros((2, ck.curve.fq_limbs), dtype=np.uint64))
    return kzg.UnivariateKzgPCS.commit(ck, t)


def link_proofs(lhs_link_hint: LinkingHint, rhs_link_hint: LinkingHint, group_layout: GroupLayout, commit_key: kzg.UnivariateProverParam) -> LinkingProof:
    """PlonkKzgSnark::link_proofs (proof_linking.rs:80-111)."""
    c = commit_key.curve
    a1, a2 = lhs_link_hint.linking_wire_poly, rhs_link_hint.linking_wire_poly
    if not (a1.is_cuda and a2.is_cuda):
        raise ValueError("linking wire polynomials must be device resident")
    diff, quotient = compute_linking_quotient(c, a1.contiguous(), a2.contiguous(), group_layout)
    quotient_commitment = _commit_dev(commit_key, quotient)
    eta = compute_quotient_challenge(c, lhs_link_hint.linking_wire_comm, rhs_link_hint.linking_wire_comm, quotient_commitment)
    # identity polynomial a_1 - a_2 - q * Z_D(eta), opened at eta (proof_linking.rs:204-221)
    z_eta = compute_vanishing_poly_eval(c, eta, group_layout)
    terms = [(1, diff)] + ([((-z_eta) % c.r, quotient)] if quotient.shape[0] else [])
    identity = poly.lincomb(c, terms)
    opening_proof, _ = kzg.UnivariateKzgPCS.open(commit_key, identity, eta)
    return LinkingProof(quotient_commitment, opening_proof)
