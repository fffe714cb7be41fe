"""GPU: PlonkKzgSnark::link_proofs on device-resident wire polynomials (mpc-jellyfish_amd/linking.py; proof_linking.rs:80-221)
against the big-int restatement (oracle/pyref_linking.py: dense long division by the expanded vanishing polynomial), and the
restated link verifier on the result -- the accept / reject cases of the reference's tests (proof_linking.rs:551-688)."""
import random

import numpy as np
import pytest

import mirror_prover as MP          # the primitive-level sequencing of the rounds: test code since round 5 (tests/mirror_prover.py)

from conftest import affine_from_limbs, build_circuit, fr_from_mont_limbs, fr_mont_limbs, verifying_key
import pyref_fs as FS

pytestmark = pytest.mark.gpu


def _pt(pc, cm):
    return None if cm.is_infinity() else affine_from_limbs(pc, cm.xy)


def _prove_linked(mj, pyref, curve_id, log_n, layout, shared, rng, srs_beta, ck):
    """One TurboPlonk proof of a circuit whose proof-linking gates hold `shared` on the layout's rows; returns
    (proof bytes, verifying-key pieces, public input, LinkingHint, prover)."""
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    n = 1 << log_n
    start, _ = layout.range_in_nth_roots(log_n)
    spacing = 1 << (log_n - layout.alignment)
    reserved = {start + i * spacing: v for i, v in enumerate(shared)}
    sel, sig, k, w, pi = build_circuit(pc, log_n, rng, reserved=reserved)
    dom = mj.Radix2EvaluationDomain(c, log_n)
    prover = MP.TurboPlonkProver(c, n, [dom.ifft(fr_mont_limbs(c, s)) for s in sel], [dom.ifft(fr_mont_limbs(c, s)) for s in sig], k, ck)
    blind = mj.snark.draw_blinders(c, mj.rng.ChaChaRng(bytes([log_n]) * 32, 12), 5, False)
    src = mj.prover.TranscriptChallenges(prover, pi[:4])
    core = prover.prove(np.stack([fr_mont_limbs(c, col) for col in w]), fr_mont_limbs(c, pi), src, blind)
    hint = mj.linking.LinkingHint(prover.last["wire_polys"][mj.linking.PROOF_LINK_WIRE_IDX].clone(), core.wires_poly_comms[0])
    return mj.snark.serialize_proof(c, core), pi[:4], hint, prover


@pytest.mark.parametrize("curve_id,log_n1,log_n2,layout_args", [(1, 6, 8, (5, 2, 6)), (0, 9, 7, (7, 5, 40)), (1, 5, 5, (5, 4, 20))])
def test_link_two_device_proofs(gpu, mj, pyref, curve_id, log_n1, log_n2, layout_args):
    import pyref_linking as L
    import pyref_verifier as V
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    r = c.r
    rng = random.Random(77 + curve_id + log_n1)
    layout = mj.linking.GroupLayout(*layout_args)
    olayout = L.GroupLayout(*layout_args)
    shared = [rng.randrange(r) for _ in range(layout.size)]
    srs_beta = rng.randrange(1, r)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, (1 << max(log_n1, log_n2)) + 2)
    G = pyref.g1_gen(pc)
    proofs = []
    for log_n in (log_n1, log_n2):
        proof_bytes, pub, hint, prover = _prove_linked(mj, pyref, curve_id, log_n, layout, shared, rng, srs_beta, ck)
        vk = verifying_key(mj, pc, prover, len(pub))
        assert V.verify(pc, FS.StandardTranscript(pc, b"PlonkProof"), vk, pub, proof_bytes, G, srs_beta), "the linked circuit's own proof"
        proofs.append((V.deserialize_proof(pc, proof_bytes), hint))
        prover.release()
    (pr1, h1), (pr2, h2) = proofs
    link = mj.linking.link_proofs(h1, h2, layout, ck)
    # the restatement on the downloaded polynomials
    ints = lambda t: fr_from_mont_limbs(c, t.cpu().numpy().view(np.uint64).reshape(-1, 4))
    a1, a2 = ints(h1.linking_wire_poly), ints(h2.linking_wire_poly)
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkLinkingProof")
    a1c, a2c = pr1["wires_poly_comms"][0], pr2["wires_poly_comms"][0]
    assert a1c == _pt(pc, h1.linking_wire_comm) and a2c == _pt(pc, h2.linking_wire_comm)
    want = L.link_proofs(pc, a1, a2, a1c, a2c, olayout, srs_beta, fresh())
    assert _pt(pc, link.quotient_commitment) == want["quotient_commitment"]
    assert _pt(pc, link.opening_proof) == want["opening_proof"]
    assert link.serialize_compressed() == L.serialize_link_proof(lambda p: FS.g1_bytes(pc, p), want["quotient_commitment"], want["opening_proof"])
    # the verifier's side: commitments out of the two serialized Plonk proofs (proof_linking.rs:240-271)
    accept = lambda lp, lay=olayout: L.verify_link_proof(pc, fresh(), a1c, a2c, _pt(pc, lp.quotient_commitment), _pt(pc, lp.opening_proof), lay, srs_beta)
    assert accept(link)
    open_key = V.open_key_for_testing(pc, srs_beta)                          # ... and as the reference's pairing equation
    assert L.verify_link_proof(pc, fresh(), a1c, a2c, _pt(pc, link.quotient_commitment), _pt(pc, link.opening_proof), olayout, None, open_key=open_key)
    al, off, size = layout_args
    assert not accept(link, lay=L.GroupLayout(al + 1, off, size)), "wrong alignment"
    assert not accept(link, lay=L.GroupLayout(al, off + 1, size)), "wrong offset"
    # linking on a layout the circuits do not share: same (remainder-dropping) quotient as the reference's division, rejected
    bad_args = (al, off - 1, size)
    bad = mj.linking.link_proofs(h1, h2, mj.linking.GroupLayout(*bad_args), ck)
    want_bad = L.link_proofs(pc, a1, a2, a1c, a2c, L.GroupLayout(*bad_args), srs_beta, fresh())
    assert _pt(pc, bad.quotient_commitment) == want_bad["quotient_commitment"] and _pt(pc, bad.opening_proof) == want_bad["opening_proof"]
    assert not accept(bad, lay=L.GroupLayout(*bad_args))
    # a proof linked with itself (proof_linking.rs:124-127): zero quotient, commitments at infinity
    same = mj.linking.link_proofs(h1, h1, layout, ck)
    assert same.quotient_commitment.is_infinity() and same.opening_proof.is_infinity()
    assert L.verify_link_proof(pc, fresh(), a1c, a1c, None, None, olayout, srs_beta)
    # a commit key too short for the wire polynomial: PCSError like UnivariateKzgPCS::commit (mod.rs:98-104)
    short = ck.trim(layout.size + 1)
    with pytest.raises(mj.PCSError):
        mj.linking.link_proofs(h1, h2, layout, short)
    with pytest.raises(ValueError):
        mj.linking.link_proofs(mj.linking.LinkingHint(h1.linking_wire_poly.cpu(), h1.linking_wire_comm), h2, layout, ck)
    ck.release()


def test_link_proofs_with_different_witnesses_rejected(gpu, mj, pyref):
    """proof_linking.rs:605-648"""
    import pyref_linking as L
    c, pc = mj.params.CURVES[1], pyref.CURVES[1]
    rng = random.Random(5)
    layout = mj.linking.GroupLayout(4, 2, 9)
    shared = [rng.randrange(c.r) for _ in range(layout.size)]
    srs_beta = rng.randrange(1, c.r)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, (1 << 7) + 2)
    _, _, h1, p1 = _prove_linked(mj, pyref, 1, 6, layout, shared, rng, srs_beta, ck)
    other = list(shared)
    other[7] = (other[7] + 1) % c.r
    _, _, h2, p2 = _prove_linked(mj, pyref, 1, 7, layout, other, rng, srs_beta, ck)
    link = mj.linking.link_proofs(h1, h2, layout, ck)
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkLinkingProof")
    args = (_pt(pc, h1.linking_wire_comm), _pt(pc, h2.linking_wire_comm), _pt(pc, link.quotient_commitment), _pt(pc, link.opening_proof))
    assert not L.verify_link_proof(pc, fresh(), *args, L.GroupLayout(4, 2, 9), srs_beta)
    sub = mj.linking.link_proofs(h1, h2, mj.linking.GroupLayout(4, 2, 7), ck)                  # the first seven values alone do link
    assert L.verify_link_proof(pc, fresh(), args[0], args[1], _pt(pc, sub.quotient_commitment), _pt(pc, sub.opening_proof), L.GroupLayout(4, 2, 7), srs_beta)
    p1.release(); p2.release(); ck.release()


@pytest.mark.parametrize("curve_id,log_n,layout_args", [(0, 16, (12, 17, 300)), (1, 20, (18, 1000, 1000)), (0, 20, (20, 5, 64))])
def test_link_large_polynomials(gpu, mj, pyref, curve_id, log_n, layout_args):
    """Masked wire polynomials of 2^16 / 2^20 (+2) coefficients that agree on a link domain of up to 1000 points: the device
    quotient is exact (q * Z_D == a_1 - a_2 at a random point) and the restated verifier accepts -- in its pairing form too."""
    import torch
    import pyref_linking as L
    import pyref_verifier as V
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    r = c.r
    rng = random.Random(11 + log_n)
    n = 1 << log_n
    layout, olayout = mj.linking.GroupLayout(*layout_args), L.GroupLayout(*layout_args)
    srs_beta = rng.randrange(1, r)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)
    dom = mj.Radix2EvaluationDomain(c, log_n)
    v1 = mj.params.random_fr_mont(c, n, seed=3)
    v2 = mj.params.random_fr_mont(c, n, seed=4)
    al, _, size = layout_args
    start, _ = layout.range_in_nth_roots(log_n)
    rows = start + (1 << (log_n - al)) * np.arange(size)
    v2[rows] = v1[rows]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()

    def masked(vals, b):                                                     # a(X) + (b_0 + b_1 X)(X^n - 1): n + 2 coefficients, same values on H
        t = dev(np.concatenate([dom.ifft(vals), np.zeros((2, 4), dtype=np.uint64)]))
        mj.poly.mask(c, [t], n, [b])
        return t

    a1, a2 = masked(v1, [rng.randrange(r), rng.randrange(r)]), masked(v2, [rng.randrange(r), rng.randrange(r)])
    cm = lambda t: mj.UnivariateKzgPCS.commit(ck, t)
    h1, h2 = mj.linking.LinkingHint(a1, cm(a1)), mj.linking.LinkingHint(a2, cm(a2))
    link = mj.linking.link_proofs(h1, h2, layout, ck)
    diff, quotient = mj.linking.compute_linking_quotient(c, a1, a2, layout)
    assert quotient.shape[0] == n + 2 - size
    x = rng.randrange(r)
    assert mj.poly.evaluate(c, quotient, x)[0] * L.vanishing_eval(pc, olayout, x) % r == mj.poly.evaluate(c, diff, x)[0]
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkLinkingProof")
    args = (_pt(pc, h1.linking_wire_comm), _pt(pc, h2.linking_wire_comm), _pt(pc, link.quotient_commitment), _pt(pc, link.opening_proof))
    assert L.verify_link_proof(pc, fresh(), *args, olayout, srs_beta)
    assert L.verify_link_proof(pc, fresh(), *args, olayout, None, open_key=V.open_key_for_testing(pc, srs_beta))
    assert not L.verify_link_proof(pc, fresh(), *args, L.GroupLayout(al, layout_args[1], size + 1), srs_beta)
    ck.release()


@pytest.mark.parametrize("curve_id", [0, 1])
def test_div_by_roots_of_unity_matches_long_division(gpu, mj, pyref, curve_id):
    """mzk_poly_div_roots_dev against the dense long division by the expanded vanishing polynomial (proof_linking.rs:119-158),
    on both of its paths: exact (p vanishes on the domain: NTT path) and with a remainder (factor-by-factor path), roots finer
    or coarser than the polynomial's length, repeated roots, and the degenerate lengths."""
    import torch
    import pyref_linking as L
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    r = c.r
    rng = random.Random(300 + curve_id)
    dev = lambda ints: torch.from_numpy(fr_mont_limbs(c, ints).view(np.int64)).cuda()
    ints = lambda t: fr_from_mont_limbs(c, t.cpu().numpy().view(np.uint64).reshape(-1, 4)) if t.shape[0] else []
    cases = [(100, 5, 3, 7), (100, 9, 500, 12), (1500, 6, 60, 50), (5000, 13, 100, 64), (64, 6, 0, 64), (777, 10, 1020, 9), (4097, 12, 0, 1)]
    for length, log_order, first, count in cases:
        lay = L.GroupLayout(log_order, first, count)
        z = L.vanishing_polynomial(pc, lay)
        s = [rng.randrange(r) for _ in range(length - count)]
        exact = [0] * length                                               # s * Z_D
        for i, x in enumerate(s):
            for j, y in enumerate(z):
                exact[i + j] = (exact[i + j] + x * y) % r
        got = ints(mj.poly.div_by_roots_of_unity(c, dev(exact), log_order, first, count))
        assert got == s, ("exact", length, log_order, first, count)
        rough = [(v + rng.randrange(r)) % r for v in exact]                 # a remainder appears
        got = ints(mj.poly.div_by_roots_of_unity(c, dev(rough), log_order, first, count))
        assert L.pstrip(got) == L.pdiv(pc, rough, z), ("with remainder", length, log_order, first, count)
    # more roots than the order of w: the factors repeat; still the floor division by their product
    p = [rng.randrange(r) for _ in range(90)]
    got = ints(mj.poly.div_by_roots_of_unity(c, dev(p), 3, 2, 11))
    assert L.pstrip(got) == L.pdiv(pc, p, L.vanishing_polynomial(pc, L.GroupLayout(3, 2, 11)))
    # degenerate lengths
    assert mj.poly.div_by_roots_of_unity(c, dev(p[:5]), 4, 0, 5).shape[0] == 0
    assert mj.poly.div_by_roots_of_unity(c, dev(p[:5]), 4, 0, 9).shape[0] == 0
    assert ints(mj.poly.div_by_roots_of_unity(c, dev(p[:5]), 4, 3, 0)) == p[:5]
    with pytest.raises(mj.MzkError):
        mj.poly.div_by_roots_of_unity(c, dev(p), 33, 0, 2)
