"""GPU: the device-resident five-round TurboPlonk prover core against the big-int restatement
(oracle/pyref_plonk.py: schoolbook polynomial arithmetic, no FFT) on a satisfied circuit with copy
constraints -- every polynomial, the 10 evaluations, and the 13 commitments (checked through the SRS
trapdoor: commit(p) == [p(beta_srs)]G)."""
import random

import numpy as np
import pytest

import mirror_prover as MP          # the primitive-level sequencing of the rounds: test code since round 5 (tests/mirror_prover.py)

from conftest import affine_from_limbs, build_circuit, fr_from_mont_limbs, fr_mont_limbs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve_id,log_n", [(0, 4), (1, 5)])
def test_prover_core_matches_bigint_restatement(gpu, mj, pyref, curve_id, log_n):
    import pyref_plonk as PP
    c = mj.params.CURVES[curve_id]
    pc = pyref.CURVES[curve_id]
    n, r = 1 << log_n, c.r
    rng = random.Random(2024 + curve_id)
    sel, sigma_vals, k, w, pi = build_circuit(pc, log_n, rng)
    blind = {"wires": [[rng.randrange(r), rng.randrange(r)] for _ in range(5)], "z": [rng.randrange(r) for _ in range(3)],
             "quot": [rng.randrange(r) for _ in range(4)]}
    ch = {x: rng.randrange(r) for x in ("beta", "gamma", "alpha", "zeta", "v")}
    srs_beta = rng.randrange(r)
    want = PP.prove_core(pc, log_n, sel, sigma_vals, k, w, pi, blind, ch, srs_beta)
    assert want["divisible"] and want["quot_degree_ok"], "the test circuit must be satisfied"

    dom = mj.Radix2EvaluationDomain(c, log_n)
    sel_polys = [dom.ifft(fr_mont_limbs(c, s)) for s in sel]
    sig_polys = [dom.ifft(fr_mont_limbs(c, s)) for s in sigma_vals]
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)          # n + 3 powers (srs.rs:88)
    prover = MP.TurboPlonkProver(c, n, sel_polys, sig_polys, k, ck)
    proof = prover.prove(np.stack([fr_mont_limbs(c, col) for col in w]), fr_mont_limbs(c, pi),
                         mj.prover.ProverChallenges(**ch), mj.prover.Blinders(blind["wires"], blind["z"], blind["quot"]))
    host = lambda t: fr_from_mont_limbs(c, t.cpu().numpy().view(np.uint64))
    strip = lambda a: PP.pstrip(a)
    for i in range(5):
        assert strip(host(prover.last["wire_polys"][i])) == strip(want["wire_polys"][i]), ("wire poly", i)
        assert strip(host(prover.last["split"][i])) == strip(want["split"][i]), ("split quotient", i)
    assert strip(host(prover.last["z_poly"])) == strip(want["z_poly"])
    assert strip(host(prover.last["quot"])) == strip(want["quot"])
    assert strip(host(prover.last["lin"])) == strip(want["lin_poly"])
    assert strip(host(prover.last["opening"])) == strip(want["opening_poly"])
    assert strip(host(prover.last["shifted"])) == strip(want["shifted_opening_poly"])
    assert proof.wires_evals == want["wires_evals"]
    assert proof.wire_sigma_evals == want["wire_sigma_evals"]
    assert proof.perm_next_eval == want["perm_next_eval"]
    G = pyref.g1_gen(pc)
    dl = want["commit_dlogs"]
    for i in range(5):
        assert affine_from_limbs(pc, proof.wires_poly_comms[i].xy) == pyref.g1_mul(pc, dl["wires"][i], G), ("wire commitment", i)
        assert affine_from_limbs(pc, proof.split_quot_poly_comms[i].xy) == pyref.g1_mul(pc, dl["split"][i], G), ("quotient commitment", i)
    assert affine_from_limbs(pc, proof.prod_perm_poly_comm.xy) == pyref.g1_mul(pc, dl["z"], G)
    assert affine_from_limbs(pc, proof.opening_proof.xy) == pyref.g1_mul(pc, dl["opening"], G)
    assert affine_from_limbs(pc, proof.shifted_opening_proof.xy) == pyref.g1_mul(pc, dl["shifted_opening"], G)
    # the KZG opening identity the verifier checks, through the trapdoor: batch(beta) - batch(zeta) = (beta - zeta) W(beta)
    # (batch(zeta) is what the verifier recomputes from the evaluations; here from the restated polynomials)
    b_at = lambda x: sum(pow(ch["v"], i, r) * pyref.poly_eval(pc, p, x) for i, p in
                         enumerate([want["lin_poly"]] + want["wire_polys"] + want["sigmas"][:4])) % r
    assert (b_at(srs_beta) - b_at(ch["zeta"])) % r == (srs_beta - ch["zeta"]) * dl["opening"] % r
    prover.release()
    ck.release()


def test_prover_with_merlin_transcript(gpu, mj, pyref):
    """Challenges derived by the Merlin transcript mirror between the rounds (snark.rs:263-431): the restated
    prover, fed the challenges the device run derived, must reproduce the same proof -- and a transcript
    replayed over the RESTATED commitments must derive the same challenges."""
    import pyref_plonk as PP
    curve_id, log_n = 0, 4
    c = mj.params.CURVES[curve_id]
    pc = pyref.CURVES[curve_id]
    n, r = 1 << log_n, c.r
    rng = random.Random(99)
    sel, sigma_vals, k, w, pi = build_circuit(pc, log_n, rng)
    blind = {"wires": [[rng.randrange(r), rng.randrange(r)] for _ in range(5)], "z": [rng.randrange(r) for _ in range(3)],
             "quot": [rng.randrange(r) for _ in range(4)]}
    srs_beta = rng.randrange(r)
    dom = mj.Radix2EvaluationDomain(c, log_n)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)
    prover = MP.TurboPlonkProver(c, n, [dom.ifft(fr_mont_limbs(c, s)) for s in sel], [dom.ifft(fr_mont_limbs(c, s)) for s in sigma_vals], k, ck)
    pub = [pi[3]]                                              # one public input (row 3)
    src = mj.prover.TranscriptChallenges(prover, pub)
    proof = prover.prove(np.stack([fr_mont_limbs(c, col) for col in w]), fr_mont_limbs(c, pi), src,
                         mj.prover.Blinders(blind["wires"], blind["z"], blind["quot"]))
    ch = {x: src.challenges[x] for x in ("beta", "gamma", "alpha", "zeta", "v")}
    assert len(set(ch.values())) == 5
    want = PP.prove_core(pc, log_n, sel, sigma_vals, k, w, pi, blind, ch, srs_beta)
    assert want["divisible"]
    G = pyref.g1_gen(pc)
    dl = want["commit_dlogs"]
    assert affine_from_limbs(pc, proof.opening_proof.xy) == pyref.g1_mul(pc, dl["opening"], G)
    assert affine_from_limbs(pc, proof.shifted_opening_proof.xy) == pyref.g1_mul(pc, dl["shifted_opening"], G)
    assert proof.wires_evals == want["wires_evals"] and proof.perm_next_eval == want["perm_next_eval"]
    # replay the transcript over the restated commitments (points from big-int scalar multiplications)
    t = mj.transcript.StandardTranscript(c)
    sel_pts = [pyref.g1_mul(pc, pyref.poly_eval(pc, p, srs_beta), G) for p in want["selectors"]]
    sig_pts = [pyref.g1_mul(pc, pyref.poly_eval(pc, p, srs_beta), G) for p in want["sigmas"]]
    t.append_vk_and_pub_input(n, 1, k, sel_pts, sig_pts, pub)
    t.append_commitments(b"witness_poly_comms", [pyref.g1_mul(pc, d, G) for d in dl["wires"]])
    t.get_and_append_challenge(b"tau")
    assert t.get_and_append_challenge(b"beta") == ch["beta"] and t.get_and_append_challenge(b"gamma") == ch["gamma"]
    t.append_commitment(b"perm_poly_comms", pyref.g1_mul(pc, dl["z"], G))
    assert t.get_and_append_challenge(b"alpha") == ch["alpha"]
    t.append_commitments(b"quot_poly_comms", [pyref.g1_mul(pc, d, G) for d in dl["split"]])
    assert t.get_and_append_challenge(b"zeta") == ch["zeta"]
    prover.release()
    ck.release()


def test_prove_from_host_resident_witness(gpu, mj):
    """A witness that starts in host memory (the reference gathers `witness[wire_variable(i, j)]` on the host,
    relation/src/constraint_system.rs:1225-1247): (i) the gathered W x n wire table as a page-locked CPU tensor, uploaded column by
    column under the wire iNTTs; (ii) snark.HostWitness -- the witness VECTOR alone, gathered on the device over the resident
    variable-index table (mzk_plonk_gather_witness_dev).  Same proof bytes as the device-resident witness, both proof systems."""
    import dataclasses
    import torch
    for curve_id, kind in ((0, "TurboPlonk"), (1, "UltraPlonk")):
        c = mj.params.CURVES[curve_id]
        cs = mj.snark.gen_circuit_for_bench(c, 300, kind)
        assert torch.equal(cs.witness[cs.wire_variables.long().reshape(-1)].reshape(cs.wire_values.shape), cs.wire_values)
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 0xabcdef, cs.n + 2)
        pk = mj.snark.preprocess(ck, cs)
        want = mj.snark.prove(mj.rng.test_rng(), cs, pk)[1]
        table = dataclasses.replace(cs, wire_values=cs.wire_values.cpu().pin_memory())
        vector = dataclasses.replace(cs, wire_values=mj.snark.HostWitness(cs.witness.cpu().pin_memory(), cs.wire_variables))
        for _ in range(2):                                             # twice: the staging buffers are reused across proofs
            assert mj.snark.prove(mj.rng.test_rng(), table, pk)[1] == want
            assert mj.snark.prove(mj.rng.test_rng(), vector, pk)[1] == want
        pk.release()
        ck.release()
