"""Aggregated proofs over several instances -- host-side mirror of `PlonkKzgSnark::batch_prove` /
`batch_prove_internal` (plonk/src/proof_system/snark.rs:64-78, 201-469) on the device stages of prover.TurboPlonkProver.

Every instance keeps its own device-resident proving key and workspace (a TurboPlonkProver); the rounds are interleaved
exactly as the reference interleaves them -- round k of every instance, then one transcript challenge -- because that order
fixes both the transcript and the order of the `prng` draws.  What the instances share:
  * one quotient polynomial  t = sum_k alpha_base_k t_k,  alpha_base_{k+1} = alpha_base_k * alpha^3 (alpha^7 with Plookup)
    (prover.rs:661-669).  The reference sums the coset evaluations before the single inverse coset FFT; both maps being linear,
    the same coefficients come from the per-instance quotients t_k (one fused kernel + one inverse coset NTT each) combined by
    one mzk_poly_lincomb_dev, split and committed with the FIRST instance's commit key (snark.rs:352-360);
  * one linearisation polynomial: the quotient part once, the non-quotient part of instance k times alpha_base_k
    (snark.rs:403-428);
  * the two opening proofs over the concatenated polynomial lists, powers of v running across instances (prover.rs:362-419).
"""
from __future__ import annotations

import struct

import numpy as np
from dataclasses import dataclass, field

from . import kzg, poly
from . import transcript as _transcript
from .prover import PLOOKUP_EVALS, Blinders, TurboPlonkProver


@dataclass
class BatchProofCore:
    """BatchProof (structs.rs:266-291): per-instance vectors, one split quotient, two opening proofs."""
    wires_poly_comms_vec: list
    prod_perm_poly_comms_vec: list
    poly_evals_vec: list                      # (wires_evals, wire_sigma_evals, perm_next_eval) per instance
    plookup_proofs_vec: list                  # None | (h_poly_comms, prod_lookup_poly_comm, evals dict) per instance
    split_quot_poly_comms: list
    opening_proof: kzg.Commitment
    shifted_opening_proof: kzg.Commitment
    challenges: dict = field(default_factory=dict)

    def __len__(self):
        return len(self.prod_perm_poly_comms_vec)


def _pt(c, cm: kzg.Commitment):
    if cm.is_infinity():
        return None
    from .params import fq_from_mont
    x, y = fq_from_mont(c, cm.xy)
    return (x, y)


def batch_prove(provers: list[TurboPlonkProver], wire_values: list, pub_input_values: list, pub_inputs: list, blinds: list[Blinders],
                quot_blinders: list[int], extra_transcript_init_msg: bytes | None = None) -> BatchProofCore:
    """provers[k]: the proving key of instance k on the device; wire_values[k] (W, n, 4) / pub_input_values[k] (n, 4): its witness
    and public-input evaluations on H (Montgomery); pub_inputs[k]: its public input as ints (for the transcript);
    blinds[k]: its masking scalars (the `quot` field is ignored); quot_blinders: the W - 1 scalars of the one split (round 3)."""
    if not provers:
        raise ValueError("zero number of circuits/proving keys")                                  # snark.rs:213-215
    if not (len(provers) == len(wire_values) == len(pub_input_values) == len(pub_inputs) == len(blinds)):
        raise ValueError("the number of circuits != the number of proving keys")                  # snark.rs:216-223
    if len({id(p) for p in provers}) != len(provers):
        # the device workspace (slab, coefficient forms, quotient) belongs to the TurboPlonkProver: two instances of one circuit
        # need two provers (preprocess twice) -- the reference's `prove_keys` may repeat because its Oracles live on the host
        raise ValueError("one TurboPlonkProver per instance: the same prover object was passed twice")
    p0 = provers[0]
    c, n, r, W = p0.curve, p0.n, p0.curve.r, p0.W
    for p in provers:
        if p.n != n:
            raise ValueError("proving key domain size %d != expected domain size %d" % (p.n, n))  # snark.rs:236-243
        if p.W != W:
            raise ValueError("inconsistent plonk circuit types")                                  # snark.rs:258-260
        if p.curve.curve_id != c.curve_id:
            raise ValueError("instances over different curves")
    K = len(provers)
    tick = lambda name, t0: None
    # transcript init (snark.rs:263-270)
    t = _transcript.StandardTranscript(c, b"PlonkProof")
    if extra_transcript_init_msg is not None:
        t.append_message(b"extra info", extra_transcript_init_msg)
    for p, pub in zip(provers, pub_inputs):
        sel, sig = p.vk_commitments()
        t.append_vk_and_pub_input(p.n, len(pub), p.k, [_pt(c, x) for x in sel], [_pt(c, x) for x in sig], pub)
    # round 1
    states, wires_vec = [], []
    for k, p in enumerate(provers):
        st, wires_comms = p._stage_round1(wire_values[k], pub_input_values[k], blinds[k], tick, pi_zero=not any(pub_inputs[k]))
        t.append_commitments(b"witness_poly_comms", [_pt(c, x) for x in wires_comms])
        states.append(st)
        wires_vec.append(wires_comms)
    tau = t.get_and_append_challenge(b"tau")
    # round 1.5
    h_vec = []
    for p, st in zip(provers, states):
        h_comms = p._stage_round1_5(st, tau, tick)
        if h_comms is not None:
            t.append_commitments(b"h_poly_comms", [_pt(c, x) for x in h_comms])
        h_vec.append(h_comms)
    beta = t.get_and_append_challenge(b"beta")
    gamma = t.get_and_append_challenge(b"gamma")
    # round 2
    z_vec = []
    for p, st in zip(provers, states):
        z_comm = p._stage_round2(st, beta, gamma, tick)
        t.append_commitment(b"perm_poly_comms", _pt(c, z_comm))
        z_vec.append(z_comm)
    # round 2.5
    pl_vec = []
    for p, st in zip(provers, states):
        pl_comm = p._stage_round2_5(st, tick)
        if pl_comm is not None:
            t.append_commitment(b"plookup_poly_comms", _pt(c, pl_comm))
        pl_vec.append(pl_comm)
    # round 3: per-instance quotients, one weighted sum, one split (prover.rs:661-673, 902-960)
    alpha = t.get_and_append_challenge(b"alpha")
    a3 = pow(alpha, 3, r)
    a7 = pow(alpha, 7, r)
    bases, base, terms = [], 1, []
    for p, st in zip(provers, states):
        terms.append((base, p._stage_quotient(st, alpha, tick)))
        bases.append(base)
        base = base * (a7 if p.ultra else a3) % r
    quot = terms[0][1] if K == 1 else poly.lincomb(c, terms)
    quot_len = poly.degree_len_async(quot[p0.W * (n + 1) + 2:])          # of the aggregated quotient, from the expected degree up (prover.rs:915-918)
    split = p0._split_quotient(quot, quot_blinders)
    split_comms = p0._commit(split)
    p0.check_quotient_degree(quot_len)
    t.append_commitments(b"quot_poly_comms", [_pt(c, x) for x in split_comms])
    # round 4 / 4.5: all ProofEvaluations first, then all PlookupEvaluations (snark.rs:365-399)
    zeta = t.get_and_append_challenge(b"zeta")
    evals_vec = []
    for p, st in zip(provers, states):
        we, se, zn, _ = p._stage_round4(st, zeta, tick)
        for e in we:
            t.append_field_elem(b"wire_evals", e)
        for e in se:
            t.append_field_elem(b"wire_sigma_evals", e)
        t.append_field_elem(b"perm_next_eval", zn)
        evals_vec.append((we, se, zn))
    for st in states:
        if st.pe is not None:
            t.append_plookup_evaluations(st.pe)
    # linearisation polynomial (snark.rs:403-428)
    lin_terms = p0._quotient_lin_terms(zeta, split)
    for p, st, b in zip(provers, states, bases):
        lin_terms += p._lin_poly_terms(st, b)
    lin = None
    for i in range(0, len(lin_terms), poly.MAX_TERMS - 1):
        chunk = lin_terms[i:i + poly.MAX_TERMS - 1]
        lin = poly.lincomb(c, chunk if lin is None else [(1, lin)] + chunk, out_len=n + 3)
    # round 5 (prover.rs:362-419)
    v = t.get_and_append_challenge(b"v")
    open_polys, shifted_polys = [lin], []
    for p, st in zip(provers, states):
        o, s = p._open_lists(st)
        open_polys += o
        shifted_polys += s
    import torch
    rem = torch.zeros((1, 4), dtype=torch.int64, device=lin.device)
    opening = p0._batched_witness(open_polys, v, zeta, rem_out=rem)
    shifted = p0._batched_witness(shifted_polys, v, zeta * p0.w_n % r)
    open_comms = p0._commit([opening, shifted])
    # the quotient identity at zeta over all instances (prover.check_quotient_identity): the guard against an unsatisfied witness
    from .params import fr_from_mont
    lin_constant = sum(p._lin_poly_constant(st, b) for p, st, b in zip(provers, states, bases)) % r
    opened = [e for p, st in zip(provers, states) for e in p._opened_evals(st)]
    p0.check_quotient_identity(fr_from_mont(c, rem.cpu().numpy().view(np.uint64))[0], lin_constant, opened, v)
    plookup_vec = [None if st.pe is None else (h, pl, st.pe) for st, h, pl in zip(states, h_vec, pl_vec)]
    return BatchProofCore(wires_vec, z_vec, evals_vec, plookup_vec, split_comms, open_comms[0], open_comms[1],
                          {"tau": tau, "beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v})


def serialize_batch_proof(curve, proof: BatchProofCore) -> bytes:
    """`BatchProof::serialize_compressed` (derive(CanonicalSerialize), structs.rs:266-291): fields in declaration order."""
    from .params import curve as _curve
    from .snark import _g1
    c = _curve(curve)
    fr = lambda x: _transcript.fr_bytes(c, x)
    g1 = lambda cm: _g1(c, cm)
    vec = lambda items, enc: struct.pack("<Q", len(items)) + b"".join(enc(x) for x in items)
    out = vec(proof.wires_poly_comms_vec, lambda comms: vec(comms, g1))
    out += vec(proof.prod_perm_poly_comms_vec, g1)
    out += vec(proof.poly_evals_vec, lambda ev: vec(ev[0], fr) + vec(ev[1], fr) + fr(ev[2]))                   # ProofEvaluations :440-450

    def plookup(pp):                                                                                           # Option<PlookupProof> :208-222
        if pp is None:
            return b"\x00"
        h, pl, evals = pp
        return b"\x01" + vec(h, g1) + g1(pl) + b"".join(fr(evals[name]) for name in PLOOKUP_EVALS)

    out += vec(proof.plookup_proofs_vec, plookup)
    out += vec(proof.split_quot_poly_comms, g1) + g1(proof.opening_proof) + g1(proof.shifted_opening_proof)
    return out
