"""oracle/pyref_linking.py -- TEST INFRASTRUCTURE ONLY.

Big-int restatement of the reference's proof-linking sub-protocol (plonk/src/proof_system/proof_linking.rs):
    GroupLayout                              relation/src/proof_linking/mod.rs:16-54
    PlonkKzgSnark::link_proofs               proof_linking.rs:80-111   (quotient :119-134, vanishing poly :137-158,
                                             Z_D(eta) :162-176, challenge :185-197, identity opening :204-221)
    PlonkKzgSnark::verify_link_proof         proof_linking.rs:240-286  (+ UnivariateKzgPCS::verify, univariate_kzg/mod.rs:194-221)
    LinkingProof::serialize_compressed       proof_linking.rs:33-39
Polynomials are coefficient lists (low degree first) of canonical ints; the quotient is the schoolbook long division by the
expanded vanishing polynomial, exactly as the reference computes it (`&diff / &vanishing_poly`, remainder dropped).
Commitments go through the trapdoor of the test SRS: commit(p) = [p(beta)]G; the pairing check of the KZG verifier
e(C - [v]G, H) == e(pi, [beta - z]H) is evaluated as  C - [v]G == (beta - z) pi  in G1.
PARITY UNPINNED by reference vectors (the reference's tests are accept/reject only, proof_linking.rs:533-688 -- mirrored in
tests/test_linking*.py).
"""
from __future__ import annotations

import pyref as P

PROOF_LINK_WIRE_IDX = 0                                    # relation/src/proof_linking/linkable_circuit.rs:23


class GroupLayout:
    """mod.rs:16-54: the group sits on the 2^alignment-th roots of unity, `size` of them starting at `offset`."""

    def __init__(self, alignment: int, offset: int, size: int):
        self.alignment, self.offset, self.size = alignment, offset, size

    def range_in_nth_roots(self, n: int):
        assert n >= self.alignment, "Group alignment must be <= n"
        spacing = 1 << (n - self.alignment)
        start = self.offset * spacing
        return start, start + max(self.size - 1, 0) * spacing

    def domain_generator(self, c) -> int:
        return c.root_of_unity(self.alignment)


def pstrip(a):
    a = list(a)
    while a and a[-1] == 0:
        a.pop()
    return a


def psub(c, a, b):
    n = max(len(a), len(b))
    return pstrip([((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % c.r for i in range(n)])


def pdiv(c, a, b):
    """quotient of the long division a / b (ark-poly DenseOrSparsePolynomial::divide_with_q_and_r), remainder dropped."""
    r = c.r
    a, b = pstrip(a), pstrip(b)
    assert b
    if len(a) < len(b):
        return []
    q = [0] * (len(a) - len(b) + 1)
    rem = list(a)
    inv_lead = pow(b[-1], -1, r)
    for i in range(len(q) - 1, -1, -1):
        cf = rem[i + len(b) - 1] * inv_lead % r
        q[i] = cf
        if cf:
            for j, bj in enumerate(b):
                rem[i + j] = (rem[i + j] - cf * bj) % r
    return pstrip(q)


def vanishing_polynomial(c, layout: GroupLayout):
    """proof_linking.rs:137-158"""
    r = c.r
    g = layout.domain_generator(c)
    root = pow(g, layout.offset, r)
    z = [1]
    for _ in range(layout.size):
        nz = [0] * (len(z) + 1)
        for i, cf in enumerate(z):                        # z * (X - root)
            nz[i] = (nz[i] - cf * root) % r
            nz[i + 1] = (nz[i + 1] + cf) % r
        z = nz
        root = root * g % r
    return z


def vanishing_eval(c, layout: GroupLayout, x: int) -> int:
    """proof_linking.rs:162-176"""
    r = c.r
    g = layout.domain_generator(c)
    root = pow(g, layout.offset, r)
    out = 1
    for _ in range(layout.size):
        out = out * ((x - root) % r) % r
        root = root * g % r
    return out


def quotient_challenge(transcript, a1_comm, a2_comm, quotient_comm) -> int:
    """proof_linking.rs:185-197; `transcript` is a fresh b"PlonkLinkingProof" transcript."""
    transcript.append_commitments(b"linking_wire_comms", [a1_comm, a2_comm])
    transcript.append_commitment(b"quotient_comm", quotient_comm)
    return transcript.get_and_append_challenge(b"eta")


def _commit(c, poly, srs_beta):
    d = P.poly_eval(c, poly, srs_beta) if poly else 0
    return P.g1_mul(c, d, P.g1_gen(c)) if d else None


def link_proofs(c, a1, a2, a1_comm, a2_comm, layout: GroupLayout, srs_beta: int, transcript):
    """proof_linking.rs:80-111.  Returns the LinkingProof {quotient_commitment, opening_proof} as affine points (None =
    infinity) together with the intermediate polynomials."""
    a1, a2 = pstrip(a1), pstrip(a2)
    quotient = [] if a1 == a2 else pdiv(c, psub(c, a1, a2), vanishing_polynomial(c, layout))          # :119-134
    quotient_comm = _commit(c, quotient, srs_beta)
    eta = quotient_challenge(transcript, a1_comm, a2_comm, quotient_comm)
    z_eta = vanishing_eval(c, layout, eta)
    identity = psub(c, psub(c, a1, a2), [cf * z_eta % c.r for cf in quotient])                        # :212-215
    witness = pdiv(c, identity, [-eta % c.r, 1]) if identity else []                                  # univariate_kzg/mod.rs:143-146
    return {"quotient_commitment": quotient_comm, "opening_proof": _commit(c, witness, srs_beta), "eta": eta,
            "quotient": quotient, "identity": identity, "witness": witness}


def verify_link_proof(c, transcript, a1_comm, a2_comm, quotient_comm, opening_proof, layout: GroupLayout, srs_beta, open_key=None) -> bool:
    """proof_linking.rs:240-286: a1_comm / a2_comm are wires_poly_comms[PROOF_LINK_WIRE_IDX] of the two Plonk proofs.
    The KZG check at point eta with value 0 is the reference's pairing equation when `open_key` ({g, h, beta_h}) is given
    (univariate_kzg/mod.rs:194-221: e(C - [v]g, h) == e(proof, beta_h - [eta]h)), else its trapdoor form C == (beta - eta) proof."""
    r = c.r
    eta = quotient_challenge(transcript, a1_comm, a2_comm, quotient_comm)
    z_eta = vanishing_eval(c, layout, eta)
    ident = P.g1_add(c, a1_comm, P.g1_neg(c, a2_comm) if a2_comm is not None else None)                # :275-286
    if quotient_comm is not None and z_eta:
        ident = P.g1_add(c, ident, P.g1_neg(c, P.g1_mul(c, z_eta, quotient_comm)))
    if open_key is not None:
        import pyref_pairing as PR
        pc = PR.PAIRINGS[c.curve_id]
        rhs_g2 = PR.ec_add(open_key["beta_h"], PR.ec_neg(pc.g2_mul(open_key["h"], eta)))
        return pc.multi_pairing_is_one([(ident, open_key["h"]), (P.g1_neg(c, opening_proof), rhs_g2)])
    rhs = P.g1_mul(c, (srs_beta - eta) % r, opening_proof) if opening_proof is not None and (srs_beta - eta) % r else None
    return ident == rhs


def serialize_link_proof(g1_bytes, quotient_comm, opening_proof) -> bytes:
    """derive(CanonicalSerialize) on LinkingProof{quotient_commitment, opening_proof: UnivariateKzgProof{proof}}"""
    return g1_bytes(quotient_comm) + g1_bytes(opening_proof)
