"""CPU: the proof-linking restatement (oracle/pyref_linking.py: proof_linking.rs:80-286) on the wire polynomials of two restated
Plonk proofs of different circuits and sizes -- accept / reject cases of the reference's own tests
(proof_linking.rs:551-688: no layout clash, different circuits, different witnesses, wrong alignment, wrong offset)."""
import random

import pytest

from conftest import build_circuit
import pyref_fs as FS


def two_linked_proofs(pyref, curve_id, layout, seed, log_n1=4, log_n2=5, tamper=None):
    """Two satisfied circuits (2^log_n1 and 2^log_n2 rows) sharing `layout.size` witness values on the link domain; returns the
    masked wire-0 polynomials of their restated proofs and the commitments, through the trapdoor."""
    import pyref_plonk as PP
    c = pyref.CURVES[curve_id]
    r = c.r
    rng = random.Random(seed)
    shared = [rng.randrange(r) for _ in range(layout.size)]
    srs_beta = rng.randrange(1, r)
    polys = []
    for which, log_n in enumerate((log_n1, log_n2)):
        start, end = layout.range_in_nth_roots(log_n)
        spacing = 1 << (log_n - layout.alignment)
        values = list(shared)
        if tamper is not None and which == 1:
            values[tamper] = (values[tamper] + 1) % r
        reserved = {start + i * spacing: v for i, v in enumerate(values)}
        assert max(reserved) == end < (1 << log_n) and 3 not in reserved
        sel, sig, k, w, pi = build_circuit(c, log_n, rng, reserved=reserved)
        blind = {"wires": [[rng.randrange(r) for _ in range(2)] for _ in range(5)], "z": [rng.randrange(r) for _ in range(3)],
                 "quot": [rng.randrange(r) for _ in range(4)]}
        ch = {x: rng.randrange(r) for x in ("beta", "gamma", "alpha", "zeta", "v")}
        out = PP.prove_core(c, log_n, sel, sig, k, w, pi, blind, ch, srs_beta)
        assert out["divisible"] and out["quot_degree_ok"], "proof-linking gates leave the circuit satisfied"
        a = out["wire_polys"][0]
        w_n = c.root_of_unity(log_n)
        for row, v in reserved.items():
            assert pyref.poly_eval(c, a, pow(w_n, row, r)) == v
        polys.append(a)
    G = pyref.g1_gen(c)
    comms = [pyref.g1_mul(c, pyref.poly_eval(c, a, srs_beta), G) for a in polys]
    return c, polys, comms, srs_beta


@pytest.mark.parametrize("curve_id", [0, 1])
def test_link_proof_restatement_accepts_and_rejects(pyref, mj, curve_id):
    import pyref_linking as L
    layout = L.GroupLayout(3, 2, 5)                                       # 5 of the 8th roots of unity, from the third one on
    c, (a1, a2), (c1, c2), srs_beta = two_linked_proofs(pyref, curve_id, layout, 99 + curve_id)
    pc = mj.params.CURVES[curve_id]
    fresh = lambda: FS.StandardTranscript(c, b"PlonkLinkingProof")
    lp = L.link_proofs(c, a1, a2, c1, c2, layout, srs_beta, fresh())
    assert len(a1) == 16 + 2 and len(a2) == 32 + 2 and len(lp["quotient"]) == len(a2) - layout.size
    # the dropped remainder is zero, i.e. the wire polynomials agree on the link domain
    z = L.vanishing_polynomial(c, layout)
    prod = [0] * (len(lp["quotient"]) + len(z) - 1)
    for i, x in enumerate(lp["quotient"]):
        for j, y in enumerate(z):
            prod[i + j] = (prod[i + j] + x * y) % c.r
    assert L.pstrip(prod) == L.psub(c, a1, a2)
    assert pyref.poly_eval(c, lp["identity"], lp["eta"]) == 0
    accept = lambda q, o, lay=layout, x=c1, y=c2: L.verify_link_proof(c, fresh(), x, y, q, o, lay, srs_beta)
    assert accept(lp["quotient_commitment"], lp["opening_proof"])
    # long division by the expanded Z_D == successive synthetic divisions by its linear factors (what the device path runs)
    g = layout.domain_generator(c)
    q = L.psub(c, a1, a2)
    for i in range(layout.size):
        q = L.pdiv(c, q, [-pow(g, layout.offset + i, c.r) % c.r, 1])
    assert q == lp["quotient"]
    # wrong alignment / wrong offset (proof_linking.rs:650-688): the same proof under another layout
    assert not accept(lp["quotient_commitment"], lp["opening_proof"], lay=L.GroupLayout(4, 2, 5))
    assert not accept(lp["quotient_commitment"], lp["opening_proof"], lay=L.GroupLayout(3, 3, 5))
    assert not accept(lp["opening_proof"], lp["quotient_commitment"])
    assert not accept(lp["quotient_commitment"], lp["opening_proof"], x=c2, y=c1)
    # linking under a layout the circuits do not share: the prover's division leaves a remainder, the verifier rejects
    bad_layout = L.GroupLayout(3, 1, 5)
    bad = L.link_proofs(c, a1, a2, c1, c2, bad_layout, srs_beta, fresh())
    assert not accept(bad["quotient_commitment"], bad["opening_proof"], lay=bad_layout)
    # a proof linked with itself: empty quotient, both commitments at infinity, accepted (proof_linking.rs:124-127)
    same = L.link_proofs(c, a1, a1, c1, c1, layout, srs_beta, fresh())
    assert same["quotient"] == [] and same["quotient_commitment"] is None and same["opening_proof"] is None
    assert accept(None, None, x=c1, y=c1)
    # the KZG check as the reference evaluates it (pairings over the OpenKey), on an accepted and a rejected link
    import pyref_verifier as V
    open_key = V.open_key_for_testing(c, srs_beta)
    assert L.verify_link_proof(c, fresh(), c1, c2, lp["quotient_commitment"], lp["opening_proof"], layout, None, open_key=open_key)
    assert not L.verify_link_proof(c, fresh(), c1, c2, lp["quotient_commitment"], lp["opening_proof"], L.GroupLayout(3, 3, 5), None, open_key=open_key)
    assert not L.verify_link_proof(c, fresh(), c1, c2, bad["quotient_commitment"], bad["opening_proof"], bad_layout, None, open_key=open_key)
    # serialized LinkingProof: two compressed G1 points
    g1 = lambda p: FS.g1_bytes(c, p)
    blob = L.serialize_link_proof(g1, lp["quotient_commitment"], lp["opening_proof"])
    assert len(blob) == 2 * (48 if curve_id == 0 else 32)


def test_link_proof_with_different_witnesses_is_rejected(pyref, mj):
    """proof_linking.rs:605-648: one linked value differs between the two circuits."""
    import pyref_linking as L
    layout = L.GroupLayout(3, 2, 5)
    c, (a1, a2), (c1, c2), srs_beta = two_linked_proofs(pyref, 1, layout, 123, tamper=4)
    pc = mj.params.CURVES[1]
    fresh = lambda: FS.StandardTranscript(c, b"PlonkLinkingProof")
    lp = L.link_proofs(c, a1, a2, c1, c2, layout, srs_beta, fresh())
    assert not L.verify_link_proof(c, fresh(), c1, c2, lp["quotient_commitment"], lp["opening_proof"], layout, srs_beta)
    # ... while the first four values alone still link
    sub = L.GroupLayout(3, 2, 4)
    lp = L.link_proofs(c, a1, a2, c1, c2, sub, srs_beta, fresh())
    assert L.verify_link_proof(c, fresh(), c1, c2, lp["quotient_commitment"], lp["opening_proof"], sub, srs_beta)


def test_group_layout_mirror(mj, pyref):
    """GroupLayout::range_in_nth_roots / get_domain_generator (relation/src/proof_linking/mod.rs:37-54) of the host mirror."""
    import pyref_linking as L
    lay = mj.linking.GroupLayout(4, 3, 6)
    assert lay.range_in_nth_roots(4) == (3, 8) and lay.range_in_nth_roots(7) == (24, 64)
    assert mj.linking.GroupLayout(4, 3, 0).range_in_nth_roots(5) == (6, 6)
    with pytest.raises(ValueError):
        lay.range_in_nth_roots(3)
    for cid in (0, 1):
        c = mj.params.CURVES[cid]
        g = lay.get_domain_generator(c)
        assert pow(g, 16, c.r) == 1 and pow(g, 8, c.r) != 1 and g == L.GroupLayout(4, 3, 6).domain_generator(pyref.CURVES[cid])
        assert mj.linking.compute_vanishing_poly_eval(c, 12345, lay) == L.vanishing_eval(pyref.CURVES[cid], L.GroupLayout(4, 3, 6), 12345)
    with pytest.raises(ValueError):
        mj.linking.GroupLayout(29, 0, 1).get_domain_generator(mj.params.CURVES[1])       # BN254 Fr two-adicity is 28
