"""CPU: the committed whole-proof vectors (tests/golden/proof_vectors.json, made by tests/golden/make_proof_golden.py from the
big-int restatements and the reference's deterministic `test_rng`) -- regenerated identically, and accepted by the restated
verifier in the reference's pairing form.  The GPU suite (test_golden_proofs_gpu.py) requires the device prover to emit exactly
these bytes."""
import importlib.util
import os

import pytest

from conftest import load_golden
import pyref_fs as FS
import pyref_rng as RNG

HERE = os.path.dirname(os.path.abspath(__file__))


def _generator():
    spec = importlib.util.spec_from_file_location("make_proof_golden", os.path.join(HERE, "golden", "make_proof_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def golden_vk(V, pc, vec):
    pt = lambda hx: V.g1_decompress(pc, bytes.fromhex(hx))
    vk = {"domain_size": vec["domain_size"], "num_inputs": 0, "k": [int(x, 16) for x in vec["k"]], "selector_comms": [pt(x) for x in vec["selector_comms"]],
          "sigma_comms": [pt(x) for x in vec["sigma_comms"]], "plookup": None}
    if vec["plookup_comms"] is not None:
        vk["plookup"] = {name: pt(x) for name, x in vec["plookup_comms"].items()}
    return vk


@pytest.mark.parametrize("index", [0, 1, 2, 3])
def test_golden_proof_vectors(pyref, mj, index):
    import pyref_verifier as V
    vec = load_golden("proof_vectors")[index]
    gen = _generator()
    assert (vec["curve"], vec["plonk_type"], vec["num_gates"], vec["range_bit_len"]) == gen.CASES[index]
    assert gen.build(*gen.CASES[index]) == vec, "tests/golden/proof_vectors.json is stale: run tests/golden/make_proof_golden.py"
    c, pc = mj.params.CURVES[vec["curve"]], pyref.CURVES[vec["curve"]]
    vk = golden_vk(V, pc, vec)
    proof = bytes.fromhex(vec["proof"])
    srs_beta = int(vec["srs_beta"], 16)
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkProof")
    assert V.verify(pc, fresh(), vk, [], proof, None, None, open_key=V.open_key_for_testing(pc, srs_beta))
    ch = V.compute_challenges(fresh(), vk, [], V.deserialize_proof(pc, proof))
    assert {name: "%x" % ch[name] for name in vec["challenges"]} == vec["challenges"]
    bad = bytearray(proof)
    bad[-40 if vec["plookup_comms"] is None else -2] ^= 1
    assert not V.verify(pc, fresh(), vk, [], bytes(bad), pyref.g1_gen(pc), srs_beta)


@pytest.mark.parametrize("index", [0, 1, 2, 3])
def test_golden_proof_vectors_over_the_reference_testing_setup(pyref, mj, index):
    """tests/golden/proof_vectors_refsetup.json (SRS of universal_setup_for_testing: g = G1::rand, snark.rs:495-517): regenerated
    identically; g lies on the curve and in the subgroup; the restated verifier accepts with open key (g, h, beta h) -- h is any
    G2 generator multiple for the pairing check, the prover never sees it."""
    import pyref_verifier as V
    vec = load_golden("proof_vectors_refsetup")[index]
    gen = _generator()
    assert gen.build(*gen.CASES[index], reference_setup=True) == vec, "tests/golden/proof_vectors_refsetup.json is stale: run tests/golden/make_proof_golden.py"
    pc = pyref.CURVES[vec["curve"]]
    g = (int(vec["srs_g"][0], 16), int(vec["srs_g"][1], 16))
    assert pyref.g1_on_curve(pc, g) and pyref.g1_mul(pc, pc.r, g) is None and g != pyref.g1_gen(pc)
    rng = RNG.test_rng()
    assert RNG.universal_setup_for_testing(pc, rng) == (int(vec["srs_beta"], 16), g)
    vk = golden_vk(V, pc, vec)
    proof = bytes.fromhex(vec["proof"])
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkProof")
    assert V.verify(pc, fresh(), vk, [], proof, g, int(vec["srs_beta"], 16))
    bad = bytearray(proof)
    bad[-40 if vec["plookup_comms"] is None else -2] ^= 1
    assert not V.verify(pc, fresh(), vk, [], bytes(bad), g, int(vec["srs_beta"], 16))


@pytest.mark.parametrize("index", [0, 1])
def test_golden_link_vectors(pyref, mj, index):
    """tests/golden/link_vectors.json: two proofs on one `test_rng` stream and their LinkingProof, all by the restatements --
    regenerated identically and accepted by the restated link verifier in its pairing form (proof_linking.rs:240-286)."""
    import pyref_linking as L
    import pyref_verifier as V
    vec = load_golden("link_vectors")[index]
    gen = _generator()
    assert gen.build_link(*gen.LINK_CASES[index]) == vec, "tests/golden/link_vectors.json is stale: run tests/golden/make_proof_golden.py"
    c, pc = mj.params.CURVES[vec["curve"]], pyref.CURVES[vec["curve"]]
    srs_beta = int(vec["srs_beta"], 16)
    wire0 = [V.deserialize_proof(pc, bytes.fromhex(p))["wires_poly_comms"][0] for p in vec["proofs"]]
    blob = bytes.fromhex(vec["link_proof"])
    half = len(blob) // 2
    q, o = V.g1_decompress(pc, blob[:half]), V.g1_decompress(pc, blob[half:])
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkLinkingProof")
    layout = L.GroupLayout(*vec["layout"])
    open_key = V.open_key_for_testing(pc, srs_beta)
    assert L.verify_link_proof(pc, fresh(), wire0[0], wire0[1], q, o, layout, None, open_key=open_key)
    assert L.quotient_challenge(fresh(), wire0[0], wire0[1], q) == int(vec["eta"], 16)
    al, off, size = vec["layout"]
    assert not L.verify_link_proof(pc, fresh(), wire0[0], wire0[1], q, o, L.GroupLayout(al, off, size + 1), srs_beta)
    assert not L.verify_link_proof(pc, fresh(), wire0[1], wire0[0], q, o, layout, srs_beta)


@pytest.mark.parametrize("index", [0, 1])
def test_golden_batch_vectors(pyref, mj, index):
    """tests/golden/batch_vectors.json: one aggregated BatchProof over several bench circuits by the restatements -- regenerated
    identically; the restated batch verifier accepts it (vks recomputed through the trapdoor from the restated circuits)."""
    import pyref_circuit as PC
    import pyref_verifier as V
    vec = load_golden("batch_vectors")[index]
    gen = _generator()
    assert gen.build_batch(*gen.BATCH_CASES[index]) == vec, "tests/golden/batch_vectors.json is stale: run tests/golden/make_proof_golden.py"
    c, pc = mj.params.CURVES[vec["curve"]], pyref.CURVES[vec["curve"]]
    ultra = vec["plonk_type"] == "UltraPlonk"
    W = 6 if ultra else 5
    n = vec["domain_size"]
    srs_beta = int(vec["srs_beta"], 16)
    k = RNG.compute_coset_representatives(pc, W, n)
    G = pyref.g1_gen(pc)
    log_n = n.bit_length() - 1
    def commit_vals(vals):
        return pyref.g1_mul(pc, pyref.poly_eval(pc, pyref.ntt_fast(pc, list(vals), log_n, 1, inverse=True), srs_beta), G) if any(vals) else None

    vks = []
    for g in vec["gates"]:
        _, _, _, sel, sigma, tables = PC.bench_circuit(pc, g, ultra, vec["range_bit_len"], k)
        vk = {"domain_size": n, "num_inputs": 0, "k": k, "selector_comms": [commit_vals(s) for s in sel], "sigma_comms": [commit_vals(s) for s in sigma],
              "plookup": None}
        if ultra:
            vk["plookup"] = {"range_table_comm": commit_vals(tables["range"]), "key_table_comm": commit_vals(tables["key"]),
                             "table_dom_sep_comm": commit_vals(tables["table_dom_sep"]), "q_dom_sep_comm": commit_vals(tables["q_dom_sep"])}
        vks.append(vk)
    pubs = [[] for _ in vec["gates"]]
    blob = bytes.fromhex(vec["batch_proof"])
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkProof")
    assert V.verify_batch_proof(pc, fresh(), vks, pubs, blob, None, None, open_key=V.open_key_for_testing(pc, srs_beta))
    assert not V.verify_batch_proof(pc, fresh(), vks[::-1], pubs, blob, G, srs_beta)


@pytest.mark.parametrize("name", ["proof_vectors", "proof_vectors_refsetup"])
def test_reference_proof_fixtures_when_present(name):
    """tests/golden/ref_proof_vectors.json = `PlonkKzgSnark::{preprocess, prove}` of the reference itself on the four golden cases
    (integration/rust/src/bin/gen_fixtures.rs); ref_proof_vectors_refsetup.json = the same over the reference's OWN
    `universal_setup_for_testing` (g = G1::rand: also pins the restated sampler).  When present: coset representatives, verifying-key
    commitments and the compressed proof BYTES of the restatements must equal the reference's."""
    import json
    path = os.path.join(HERE, "golden", "ref_%s.json" % name)
    if not os.path.exists(path):
        pytest.skip("reference fixtures absent (integration/rust has not been run): parity unpinned")
    ref = json.load(open(path))
    ours = load_golden(name)
    assert len(ref) == len(ours)
    for a, b in zip(ours, ref):
        for f in ("curve", "plonk_type", "num_gates", "domain_size", "srs_beta") + (("srs_g",) if "srs_g" in a else ()):
            assert a[f] == b[f]
        assert [int(x, 16) for x in a["k"]] == [int(x, 16) for x in b["k"]]
        assert a["selector_comms"] == b["selector_comms"] and a["sigma_comms"] == b["sigma_comms"]
        assert a["proof"] == b["proof"], "proof bytes differ from the reference's"
