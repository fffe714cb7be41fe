"""GPU: the device prover emits, byte for byte, the committed whole-proof vectors (tests/golden/proof_vectors.json: bench
circuit, `test_rng` blinders, Merlin transcript, by the big-int restatements alone) -- proof and verifying-key commitments."""
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("index", [0, 1, 2, 3])
def test_device_prover_reproduces_the_golden_proof(gpu, mj, index):
    vec = load_golden("proof_vectors")[index]
    c = mj.params.CURVES[vec["curve"]]
    cs = mj.snark.gen_circuit_for_bench(c, vec["num_gates"], vec["plonk_type"], range_bit_len=vec["range_bit_len"])
    assert cs.n == vec["domain_size"] and ["%x" % x for x in cs.k] == vec["k"]
    rng = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng)
    assert "%x" % srs_beta == vec["srs_beta"]
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, cs.n + 2)
    pk = mj.snark.preprocess(ck, cs)
    sel, sig = pk.vk_commitments()
    assert [mj.snark._g1(c, x).hex() for x in sel] == vec["selector_comms"] and [mj.snark._g1(c, x).hex() for x in sig] == vec["sigma_comms"]
    if pk.ultra:
        names = ("range_table_comm", "key_table_comm", "table_dom_sep_comm", "q_dom_sep_comm")
        assert dict(zip(names, [mj.snark._g1(c, x).hex() for x in pk.plookup_vk_commitments()])) == vec["plookup_comms"]
    _, proof_bytes = mj.snark.prove(rng, cs, pk)
    assert proof_bytes.hex() == vec["proof"]
    assert {name: "%x" % v for name, v in pk.last_challenges.items()} == vec["challenges"]
    pk.release()
    ck.release()


def _rng_after_setup(mj, c, reference_setup):
    """test_rng advanced past the SRS draws (beta, or universal_setup_for_testing's beta, g, h)"""
    rng = mj.rng.test_rng()
    if reference_setup:
        mj.rng.universal_setup_for_testing(c, rng)
    else:
        mj.rng.fr_rand(c, rng)
    return rng


@pytest.mark.parametrize("name,index", [("proof_vectors", 0), ("proof_vectors", 1), ("proof_vectors", 2), ("proof_vectors", 3), ("proof_vectors_refsetup", 0),
                                        ("proof_vectors_refsetup", 3)])
def test_golden_proofs_with_round_1_committed_over_the_lagrange_basis(gpu, mj, name, index):
    """TurboPlonkProver.lagrange_ck: the wire commitments of round 1 are MSMs of the wire VALUES (and the two blinders) over
    [L_i(beta)]g, [X^j Z_H(beta)]g -- the same group elements, hence the same proof bytes, as the golden vectors made from coefficient
    forms (both testing set-ups: g the generator, g = G1::rand)."""
    vec = load_golden(name)[index]
    c = mj.params.CURVES[vec["curve"]]
    cs = mj.snark.gen_circuit_for_bench(c, vec["num_gates"], vec["plonk_type"], range_bit_len=vec["range_bit_len"])
    rng = mj.rng.test_rng()
    if "srs_g" in vec:
        srs_beta, g = mj.rng.universal_setup_for_testing(c, rng)
    else:
        srs_beta, g = mj.rng.fr_rand(c, rng), None
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, cs.n + 2, g=g)
    for mode in ("from the SRS", "from beta", "off"):
        pk = mj.snark.preprocess(ck, cs, lagrange=mode == "from the SRS")           # (None, the default: from 2^13 gates on)
        assert (pk.lagrange_ck is not None) == (mode == "from the SRS")
        if mode == "from beta":
            pk.lagrange_ck = mj.UnivariateProverParam.gen_lagrange_srs_for_testing(c, srs_beta, cs.n, g=g)
        _, proof_bytes = mj.snark.prove(_rng_after_setup(mj, c, "srs_g" in vec), cs, pk)
        assert proof_bytes.hex() == vec["proof"], mode
        pk.release()
    ck.release()


@pytest.mark.parametrize("index", [0, 1])
def test_device_link_proofs_reproduce_the_golden_link(gpu, mj, index):
    """prove_with_link_hint twice + link_proofs on the device: both proofs and the LinkingProof of tests/golden/link_vectors.json."""
    vec = load_golden("link_vectors")[index]
    c = mj.params.CURVES[vec["curve"]]
    circuits = [mj.snark.gen_circuit_for_bench(c, g, "TurboPlonk") for g in vec["gates"]]
    rng = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng)
    assert "%x" % srs_beta == vec["srs_beta"]
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, circuits[0].n + 2)
    pks = [mj.snark.preprocess(ck, cs) for cs in circuits]
    hints = []
    for cs, pk, want in zip(circuits, pks, vec["proofs"]):
        _, proof_bytes, hint = mj.snark.prove_with_link_hint(rng, cs, pk)
        assert proof_bytes.hex() == want
        hints.append(hint)
    link = mj.linking.link_proofs(hints[0], hints[1], mj.linking.GroupLayout(*vec["layout"]), ck)
    assert link.serialize_compressed().hex() == vec["link_proof"]
    for pk in pks:
        pk.release()
    ck.release()


def test_cpp_host_reproduces_the_golden_proofs(gpu):
    """The compiled host (mzk_prove) against the same CPU-made vectors: proofs, verifying keys and the LinkingProof."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binp = os.path.join(root, "mpc-jellyfish_amd", "mzk_prove")
    if not os.path.exists(binp):
        subprocess.check_call(["make", "-C", os.path.join(root, "mpc-jellyfish_amd", "host"), "-s"])
    run = lambda args: json.loads(subprocess.run([binp] + [str(a) for a in args], capture_output=True, text=True, timeout=600, check=True).stdout.strip().splitlines()[-1])
    for vec in load_golden("proof_vectors"):
        got = run([vec["curve"], "ultra" if vec["plonk_type"] == "UltraPlonk" else "turbo", vec["num_gates"], 0, vec["range_bit_len"]])
        assert got["proof_hex"] == vec["proof"], (vec["curve"], vec["plonk_type"])
        assert got["vk_hex"] == "".join(vec["selector_comms"] + vec["sigma_comms"])
    for vec in load_golden("link_vectors"):
        got = run([vec["curve"], "link"] + vec["gates"] + vec["layout"])
        assert [got["proof1_hex"], got["proof2_hex"]] == vec["proofs"] and got["link_proof_hex"] == vec["link_proof"]


@pytest.mark.parametrize("index", [0, 1])
def test_device_and_cpp_batch_prove_reproduce_the_golden_batch_proof(gpu, mj, index):
    import json
    import os
    import subprocess
    vec = load_golden("batch_vectors")[index]
    c = mj.params.CURVES[vec["curve"]]
    circuits = [mj.snark.gen_circuit_for_bench(c, g, vec["plonk_type"], range_bit_len=vec["range_bit_len"]) for g in vec["gates"]]
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), circuits[0].n + 2)
    pks = [mj.snark.preprocess(ck, cs) for cs in circuits]
    core, blob = mj.snark.batch_prove(rng, circuits, pks)
    assert blob.hex() == vec["batch_proof"]
    assert {name: "%x" % v for name, v in core.challenges.items()} == vec["challenges"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([os.path.join(root, "mpc-jellyfish_amd", "mzk_prove"), str(vec["curve"]), "batch", "ultra" if vec["plonk_type"] == "UltraPlonk" else "turbo",
                          str(vec["range_bit_len"])] + [str(g) for g in vec["gates"]], capture_output=True, text=True, timeout=600, check=True)
    assert json.loads(out.stdout.strip().splitlines()[-1])["batch_proof_hex"] == vec["batch_proof"]
    for pk in pks:
        pk.release()
    ck.release()


@pytest.mark.parametrize("index", [0, 1, 2, 3])
def test_device_prover_over_the_reference_testing_setup(gpu, mj, index):
    """tests/golden/proof_vectors_refsetup.json: the same four proofs over the SRS of `universal_setup_for_testing`
    (plonk/src/proof_system/snark.rs:495-517: beta = Fr::rand, g = G1::rand, h = G2::rand from the rng `prove` continues on) --
    what plonk/benches/bench.rs and the reference's own tests prove over.  The product mirrors the three draws (rng.py), builds
    [beta^i g] on the device (mzk_srs_generate_for_testing_g) and must emit the oracle's bytes; integration/rust/gen_fixtures
    writes the reference's bytes for the same case (tests/golden/ref_proof_vectors_refsetup.json) the day it runs."""
    vec = load_golden("proof_vectors_refsetup")[index]
    assert vec["setup"] == "universal_setup_for_testing"
    c = mj.params.CURVES[vec["curve"]]
    cs = mj.snark.gen_circuit_for_bench(c, vec["num_gates"], vec["plonk_type"], range_bit_len=vec["range_bit_len"])
    rng = mj.rng.test_rng()
    srs_beta, g = mj.rng.universal_setup_for_testing(c, rng)
    assert "%x" % srs_beta == vec["srs_beta"] and ["%x" % g[0], "%x" % g[1]] == vec["srs_g"]
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, cs.n + 2, g=g)
    pk = mj.snark.preprocess(ck, cs)
    sel, sig = pk.vk_commitments()
    assert [mj.snark._g1(c, x).hex() for x in sel] == vec["selector_comms"] and [mj.snark._g1(c, x).hex() for x in sig] == vec["sigma_comms"]
    _, proof_bytes = mj.snark.prove(rng, cs, pk)
    assert proof_bytes.hex() == vec["proof"]
    pk.release()
    ck.release()
