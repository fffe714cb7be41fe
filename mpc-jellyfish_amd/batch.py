"""Aggregated proofs over several instances -- `PlonkKzgSnark::batch_prove` / `batch_prove_internal`
(plonk/src/proof_system/snark.rs:64-78, 201-469) over K handles of the round-level C ABI (prover.TurboPlonkProver).

Every instance keeps its own device-resident proving key and workspace; the rounds are interleaved exactly as the reference
interleaves them -- round k of every instance, then one transcript challenge -- because that order fixes both the transcript and the
order of the `prng` draws: rounds 1 - 2.5 and 4 per instance, rounds 3 and 5 ONCE over all handles (mzk_prover_round3 / round5 take
the K instances: one quotient t = sum_k alpha_base_k t_k with alpha_base_{k+1} = alpha_base_k alpha^3 (alpha^7 with Plookup,
prover.rs:661-669), split and committed with the first instance's commit key (snark.rs:352-360); one linearisation polynomial
(snark.rs:403-428); two opening proofs over the concatenated polynomial lists (prover.rs:362-419)).
"""
from __future__ import annotations

import struct

import numpy as np
from dataclasses import dataclass, field

from . import kzg
from . import transcript as _transcript
from . import prover as _prover
from .prover import PLOOKUP_EVALS


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


def batch_prove(provers, wire_values: list, pub_inputs: list, blinds: list, quot_blinders: list, extra_transcript_init_msg: bytes | None = None):
    """PlonkKzgSnark::batch_prove (snark.rs:64-78, 201-469) over K native handles: rounds 1 - 2.5 and 4 per instance, rounds 3 and 5
    once over all handles.  pub_inputs[k]: instance k's public input as ints, on rows 0.. of its circuit.  Returns batch.BatchProofCore."""
    if not provers:
        raise ValueError("zero number of circuits/proving keys")
    if not (len(provers) == len(wire_values) == len(pub_inputs) == len(blinds)):
        raise ValueError("the number of circuits != the number of proving keys")
    if len({id(p) for p in provers}) != len(provers):
        # the device workspace belongs to the handle: two instances of one circuit need two provers (preprocess twice) -- the reference's
        # `prove_keys` may repeat because its Oracles live on the host
        raise ValueError("one TurboPlonkProver per instance: the same prover object was passed twice")
    p0 = provers[0]
    c = p0.curve
    for p in provers:
        if p.n != p0.n:
            raise ValueError("proving key domain size %d != expected domain size %d" % (p.n, p0.n))  # snark.rs:236-243
        if p.W != p0.W:
            raise ValueError("inconsistent plonk circuit types")                                      # snark.rs:258-260
        if p.curve.curve_id != c.curve_id:
            raise ValueError("instances over different curves")
    pt = lambda cm: _pt(c, cm)
    t = _transcript.StandardTranscript(c, b"PlonkProof")
    if extra_transcript_init_msg is not None:
        t.append_message(b"extra info", extra_transcript_init_msg)
    for p, pub in zip(provers, pub_inputs):
        sel, sig = p.vk_commitments()
        t.append_vk_and_pub_input(p.n, len(pub), p.k, [pt(x) for x in sel], [pt(x) for x in sig], pub)
    wires_vec = []
    for k, p in enumerate(provers):
        wires_vec.append(p.round1(wire_values[k], list(pub_inputs[k]), blinds[k].wires))
        t.append_commitments(b"witness_poly_comms", [pt(x) for x in wires_vec[-1]])
    tau = t.get_and_append_challenge(b"tau")
    h_vec = []
    for k, p in enumerate(provers):
        h_vec.append(p.round1_5(tau, blinds[k].h) if p.ultra else None)
        if h_vec[-1] is not None:
            t.append_commitments(b"h_poly_comms", [pt(x) for x in h_vec[-1]])
    beta = t.get_and_append_challenge(b"beta")
    gamma = t.get_and_append_challenge(b"gamma")
    z_vec = []
    for k, p in enumerate(provers):
        z_vec.append(p.round2(beta, gamma, blinds[k].z))
        t.append_commitment(b"perm_poly_comms", pt(z_vec[-1]))
    pl_vec = []
    for k, p in enumerate(provers):
        pl_vec.append(p.round2_5(blinds[k].prod_lookup) if p.ultra else None)
        if pl_vec[-1] is not None:
            t.append_commitment(b"plookup_poly_comms", pt(pl_vec[-1]))
    alpha = t.get_and_append_challenge(b"alpha")
    split_comms = _prover.round3(provers, alpha, quot_blinders)
    t.append_commitments(b"quot_poly_comms", [pt(x) for x in split_comms])
    zeta = t.get_and_append_challenge(b"zeta")
    evals_vec, pes = [], []
    for p in provers:
        we, se, zn, pe = p.round4(zeta)
        for e in we:
            t.append_field_elem(b"wire_evals", e)
        for e in se:
            t.append_field_elem(b"wire_sigma_evals", e)
        t.append_field_elem(b"perm_next_eval", zn)
        evals_vec.append((we, se, zn))
        pes.append(pe)
    for pe in pes:
        if pe is not None:
            t.append_plookup_evaluations(pe)
    v = t.get_and_append_challenge(b"v")
    open_comms = _prover.round5(provers, v)
    plookup_vec = [None if pe is None else (h, pl, pe) for pe, h, pl in zip(pes, h_vec, pl_vec)]
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
