"""GPU: the PlonkKzgSnark::prove mirror on the reference's own benchmark circuit (plonk/benches/bench.rs:29-46) --
vectorised circuit builder vs the loop restatement, blinders from the ChaCha `test_rng`, Merlin transcript, proof bytes."""
import struct

import numpy as np
import pytest

from conftest import affine_from_limbs, fr_from_mont_limbs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve_id,plonk_type,num_gates,range_bits", [(0, "TurboPlonk", 32, 8), (1, "TurboPlonk", 64, 8), (1, "UltraPlonk", 32, 3),
                                                                      (0, "UltraPlonk", 16, 5)])
def test_bench_circuit_and_proof(gpu, mj, pyref, curve_id, plonk_type, num_gates, range_bits):
    import pyref_circuit as PCirc
    import pyref_plonk as PP
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    r = c.r
    ultra = plonk_type == "UltraPlonk"
    W = 6 if ultra else 5
    cs = mj.snark.gen_circuit_for_bench(c, num_gates, plonk_type, range_bit_len=range_bits)
    k = mj.rng.compute_coset_representatives(c, W, cs.n)
    n, wires, witness, sel, sigma, tables = PCirc.bench_circuit(pc, num_gates, ultra, range_bits, k)
    assert cs.n == n and cs.k == k
    host = lambda t: fr_from_mont_limbs(c, t.cpu().numpy().view(np.uint64).reshape(-1, 4))
    w_vals = [[witness[v] for v in wires[i]] for i in range(W)]
    assert host(cs.wire_values) == [x for row in w_vals for x in row]
    assert host(cs.selector_values) == [x for row in sel for x in row]
    assert host(cs.sigma_values) == [x for row in sigma for x in row]
    if ultra:
        assert host(cs.table_values) == tables["range"] + tables["key"] + tables["table_dom_sep"] + tables["q_dom_sep"]
    # preprocess + prove with the reference's deterministic randomness
    rng = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)
    pk = mj.snark.preprocess(ck, cs)
    rng_copy = mj.rng.ChaChaRng(mj.rng.TEST_RNG_SEED, 12)
    mj.rng.fr_rand(c, rng_copy)
    bl = mj.snark.draw_blinders(c, rng_copy, W, ultra)
    core, proof_bytes = mj.snark.prove(rng, cs, pk)
    # restated prover with the same blinders and the challenges the transcript produced
    src_ch = None
    blind = {"wires": bl.wires, "z": bl.z, "quot": bl.quot, "h": bl.h, "prod_lookup": bl.prod_lookup}
    # the transcript is replayed by a second device run to read the challenges (deterministic)
    src = mj.prover.TranscriptChallenges(pk, [])
    core2 = pk.prove(cs.wire_values, cs.pub_input_values, src, bl)
    ch = dict(src.challenges)
    want = PP.prove_core(pc, n.bit_length() - 1, sel, sigma, k, w_vals, [0] * n, blind, ch, srs_beta, plookup=tables)
    assert want["divisible"] and want["quot_degree_ok"], "the bench circuit's witness satisfies the circuit"
    G = pyref.g1_gen(pc)
    pt = lambda cm: affine_from_limbs(pc, cm.xy)
    dl = want["commit_dlogs"]
    assert [pt(x) for x in core.wires_poly_comms] == [pyref.g1_mul(pc, d, G) for d in dl["wires"]]
    assert [pt(x) for x in core.split_quot_poly_comms] == [pyref.g1_mul(pc, d, G) for d in dl["split"]]
    assert pt(core.opening_proof) == pyref.g1_mul(pc, dl["opening"], G) and pt(core.shifted_opening_proof) == pyref.g1_mul(pc, dl["shifted_opening"], G)
    assert core.wires_evals == want["wires_evals"] and core.perm_next_eval == want["perm_next_eval"]
    assert mj.snark.serialize_proof(c, core2) == proof_bytes, "the proof is a deterministic function of (rng, circuit, pk)"
    # proof bytes: layout of Proof::serialize_compressed (structs.rs:59-84)
    g1_len = 48 if curve_id == 0 else 32
    want_len = (8 + W * g1_len) + g1_len + (8 + W * g1_len) + 2 * g1_len + (8 + W * 32) + (8 + (W - 1) * 32) + 32 + 1
    if ultra:
        want_len += (8 + 2 * g1_len) + g1_len + 15 * 32
    assert len(proof_bytes) == want_len
    assert struct.unpack_from("<Q", proof_bytes, 0)[0] == W
    first = mj.transcript.g1_bytes(c, pt(core.wires_poly_comms[0]))
    assert proof_bytes[8:8 + g1_len] == first
    off = (8 + W * g1_len) + g1_len + (8 + W * g1_len) + 2 * g1_len + 8
    assert int.from_bytes(proof_bytes[off:off + 32], "little") == want["wires_evals"][0]
    assert proof_bytes[off + W * 32 + 8 + (W - 1) * 32 + 32] == (1 if ultra else 0)
    pk.release()
    ck.release()


@pytest.mark.parametrize("curve_id,num_gates", [(0, 1 << 10), (1, 1 << 12)])
def test_prove_at_config_c1_size_against_the_cpu_restatement(gpu, mj, cref, curve_id, num_gates):
    """BASELINE.json configs[0]: TurboPlonk over BLS12-381 at 2^10 constraints -- the CPU path there is the C restatement of the
    ark-poly / ark-ec algorithms (oracle/cref_prover.py: FFT-based quotient, serial grand product, ark-style Pippenger), fed the
    challenges the device run squeezed from its transcript.  Every commitment and evaluation of the proof must agree."""
    import cref_prover
    c = mj.params.CURVES[curve_id]
    cs = mj.snark.gen_circuit_for_bench(c, num_gates, "TurboPlonk")
    n, log_n = cs.n, cs.n.bit_length() - 1
    rng = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)
    pk = mj.snark.preprocess(ck, cs)
    bl = mj.snark.draw_blinders(c, rng, 5, False)
    src = mj.prover.TranscriptChallenges(pk, [])
    core = pk.prove(cs.wire_values, cs.pub_input_values, src, bl)
    host = lambda t: t.cpu().numpy().view(np.uint64)
    want = cref_prover.prove_turbo(curve_id, c.r, c.fr_generator, log_n, host(cs.selector_values), host(cs.sigma_values), cs.k, host(cs.wire_values),
                                   host(cs.pub_input_values), {"wires": bl.wires, "z": bl.z, "quot": bl.quot}, dict(src.challenges),
                                   ck.powers_of_g(), threads=8)
    for got, exp in zip(core.wires_poly_comms + [core.prod_perm_poly_comm] + core.split_quot_poly_comms + [core.opening_proof, core.shifted_opening_proof],
                        want["wires_comms"] + [want["z_comm"]] + want["split_comms"] + [want["opening"], want["shifted"]]):
        assert np.array_equal(got.xy, exp)
    assert core.wires_evals == want["wires_evals"] and core.wire_sigma_evals == want["wire_sigma_evals"] and core.perm_next_eval == want["perm_next_eval"]
    pk.release()
    ck.release()


@pytest.mark.parametrize("curve_id,num_gates", [(1, 1 << 10), (0, 1 << 11)])
def test_ultra_prove_against_the_cpu_restatement(gpu, mj, cref, curve_id, num_gates):
    """UltraPlonk at 2^10 / 2^11 gates (every NTT multi-pass, every MSM on the table path): the device proof against the C
    restatement of the Plookup builders, the Ultra quotient closure and the rest (oracle/cref_prover.py::prove_ultra)."""
    import cref_prover
    c = mj.params.CURVES[curve_id]
    cs = mj.snark.gen_circuit_for_bench(c, num_gates, "UltraPlonk")
    n, log_n = cs.n, cs.n.bit_length() - 1
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), n + 2)
    pk = mj.snark.preprocess(ck, cs)
    bl = mj.snark.draw_blinders(c, rng, 6, True)
    src = mj.prover.TranscriptChallenges(pk, [])
    core = pk.prove(cs.wire_values, cs.pub_input_values, src, bl)
    host = lambda t: t.cpu().numpy().view(np.uint64)
    want = cref_prover.prove_ultra(curve_id, c.r, c.fr_generator, log_n, host(cs.selector_values), host(cs.sigma_values), host(cs.table_values), cs.k,
                                   host(cs.wire_values), host(cs.pub_input_values),
                                   {"wires": bl.wires, "z": bl.z, "quot": bl.quot, "h": bl.h, "prod_lookup": bl.prod_lookup}, dict(src.challenges),
                                   ck.powers_of_g(), threads=8)
    got = core.wires_poly_comms + core.h_poly_comms + [core.prod_perm_poly_comm, core.prod_lookup_poly_comm] + core.split_quot_poly_comms + \
        [core.opening_proof, core.shifted_opening_proof]
    exp = want["wires_comms"] + want["h_comms"] + [want["z_comm"], want["prod_lookup_comm"]] + want["split_comms"] + [want["opening"], want["shifted"]]
    for i, (g, e) in enumerate(zip(got, exp)):
        assert np.array_equal(g.xy, e), i
    assert core.wires_evals == want["wires_evals"] and core.wire_sigma_evals == want["wire_sigma_evals"] and core.perm_next_eval == want["perm_next_eval"]
    assert core.plookup_evals == want["plookup_evals"]
    pk.release()
    ck.release()
