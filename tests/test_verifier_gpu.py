"""GPU: proofs made by the device prover are ACCEPTED by the restated reference verifier (oracle/pyref_verifier.py:
Proof::deserialize_compressed, Verifier::compute_challenges / prepare_pcs_info / batch_verify_opening_proofs,
verifier.rs:68-733) -- from the serialized proof bytes alone, with the challenges recomputed from the transcript -- and
rejected once a byte changes; the final check both as the reference's pairing product (oracle/pyref_pairing.py) and in its
trapdoor form.  The verifier shares no code with the device path or with the restated provers."""
import random

import numpy as np
import pytest

from conftest import build_circuit, build_ultra_circuit, fr_mont_limbs, verifying_key
import pyref_fs as FS

pytestmark = pytest.mark.gpu


def rejects(V, *args, **kw):
    """False from the final check, or a proof that does not even deserialize."""
    try:
        return not V.verify(*args, **kw)
    except V.VerifyError:
        return True


@pytest.mark.parametrize("curve_id,plonk_type,num_gates,range_bits", [(0, "TurboPlonk", 1 << 12, 8), (1, "TurboPlonk", 100, 8),
                                                                      (1, "UltraPlonk", 1 << 11, 8), (0, "UltraPlonk", 40, 4),
                                                                      (1, "TurboPlonk", 1 << 16, 8), (0, "UltraPlonk", 1 << 17, 8),
                                                                      (0, "TurboPlonk", 1 << 20, 8), (1, "UltraPlonk", 1 << 22, 8)])
def test_bench_circuit_proof_verifies(gpu, mj, pyref, curve_id, plonk_type, num_gates, range_bits):
    """PlonkKzgSnark::prove on the reference's bench circuit (plonk/benches/bench.rs:29-46), then PlonkKzgSnark::verify -- up to
    BASELINE.json's configurations C4 (TurboPlonk, BLS12-381, 2^20 gates) and C5 (UltraPlonk, BN254, 2^22 gates) (the verifier's work does not grow with the circuit: ~50 G1 scalar multiplications)."""
    import pyref_verifier as V
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    cs = mj.snark.gen_circuit_for_bench(c, num_gates, plonk_type, range_bit_len=range_bits)
    rng = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, cs.n + 2)
    pk = mj.snark.preprocess(ck, cs)
    vk = verifying_key(mj, pc, pk, 0)
    G = pyref.g1_gen(pc)
    for extra in (None, b"extra message"):
        _, proof_bytes = mj.snark.prove(rng, cs, pk, extra_transcript_init_msg=extra)
        fresh = lambda: FS.StandardTranscript(pc, b"PlonkProof")
        assert V.verify(pc, fresh(), vk, [], proof_bytes, G, srs_beta, extra_msg=extra)
        assert not V.verify(pc, fresh(), vk, [], proof_bytes, G, srs_beta, extra_msg=b"another message")
    # the final step as the reference evaluates it: the product of two pairings over the OpenKey, no trapdoor (verifier.rs:226-250)
    open_key = V.open_key_for_testing(pc, srs_beta)
    assert V.verify(pc, fresh(), vk, [], proof_bytes, None, None, extra_msg=extra, open_key=open_key)
    assert not V.verify(pc, fresh(), vk, [], proof_bytes, None, None, extra_msg=extra, open_key=V.open_key_for_testing(pc, srs_beta + 1))
    # any altered scalar in the proof is rejected (evaluations sit behind the 3W + 3 commitments)
    g1_len = 48 if curve_id == 0 else 32
    W = cs.num_wire_types
    off = (8 + W * g1_len) + g1_len + (8 + W * g1_len) + 2 * g1_len + 8
    for at in (off, off + 32 * (W - 1), off + W * 32 + 8 + 3 * 32, len(proof_bytes) - (2 if pk.ultra else 40)):
        bad = bytearray(proof_bytes)
        bad[at] ^= 1
        assert rejects(V, pc, fresh(), vk, [], bytes(bad), G, srs_beta, extra_msg=extra), at
    # a verifying key of another circuit does not accept it
    vk2 = dict(vk, sigma_comms=vk["sigma_comms"][1:] + vk["sigma_comms"][:1])
    assert not V.verify(pc, fresh(), vk2, [], proof_bytes, G, srs_beta, extra_msg=extra)
    pk.release()
    ck.release()


@pytest.mark.parametrize("curve_id,ultra,log_n", [(0, False, 6), (1, False, 9), (1, True, 6), (0, True, 8)])
def test_proof_with_public_input_and_copy_constraints_verifies(gpu, mj, pyref, curve_id, ultra, log_n):
    """A circuit with a non-zero public input, copy constraints over all wires and (Ultra) key/range lookups."""
    import pyref_verifier as V
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    n, r = 1 << log_n, c.r
    rng = random.Random(4100 + curve_id + 2 * ultra)
    W = 6 if ultra else 5
    plookup = None
    if ultra:
        sel, sig, k, w, pi, tabs = build_ultra_circuit(pc, log_n, rng)
        plookup = tabs
    else:
        sel, sig, k, w, pi = build_circuit(pc, log_n, rng)
    dom = mj.Radix2EvaluationDomain(c, log_n)
    srs_beta = rng.randrange(1, r)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)
    kw = {"plookup": {name: dom.ifft(fr_mont_limbs(c, plookup[key])) for name, key in
                      zip(mj.plonk.PLOOKUP_TABLE_POLYS, ("range", "key", "table_dom_sep", "q_dom_sep"))}} if ultra else {}
    prover = mj.prover.TurboPlonkProver(c, n, [dom.ifft(fr_mont_limbs(c, s)) for s in sel], [dom.ifft(fr_mont_limbs(c, s)) for s in sig], k, ck, **kw)
    pub = pi[:4]
    assert pub[3] != 0 and not any(pi[4:])
    blind = mj.snark.draw_blinders(c, mj.rng.test_rng(), W, ultra)
    src = mj.prover.TranscriptChallenges(prover, pub)
    core = prover.prove(np.stack([fr_mont_limbs(c, col) for col in w]), fr_mont_limbs(c, pi), src, blind)
    proof_bytes = mj.snark.serialize_proof(c, core)
    vk = verifying_key(mj, pc, prover, len(pub))
    G = pyref.g1_gen(pc)
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkProof")
    assert V.verify(pc, fresh(), vk, pub, proof_bytes, G, srs_beta)
    # the verifier's transcript reproduces the prover's challenges (and draws u after the opening proofs)
    pr = V.deserialize_proof(pc, proof_bytes)
    ch = V.compute_challenges(fresh(), vk, pub, pr)
    assert {x: ch[x] for x in src.challenges} == src.challenges and ch["u"] not in src.challenges.values()
    # another public input is rejected (it changes the transcript AND the PI polynomial); so is one forced past the transcript
    assert not V.verify(pc, fresh(), vk, pub[:3] + [(pub[3] + 1) % r], proof_bytes, G, srs_beta)
    info = V.prepare_pcs_info(pc, vk, pub[:3] + [(pub[3] + 1) % r], pr, ch)
    assert not V.batch_verify_opening_proof(pc, G, srs_beta, info)
    assert V.batch_verify_opening_proof(pc, G, srs_beta, V.prepare_pcs_info(pc, vk, pub, pr, ch))
    assert not V.verify(pc, fresh(), vk, pub, proof_bytes, G, (srs_beta + 1) % r), "another SRS"
    prover.release()
    ck.release()
