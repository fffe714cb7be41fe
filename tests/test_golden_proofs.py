"""CPU: the committed whole-proof vectors (tests/golden/proof_vectors.json, made by tests/golden/make_proof_golden.py from the
big-int restatements and the reference's deterministic `test_rng`) -- regenerated identically, and accepted by the restated
verifier in the reference's pairing form.  The GPU suite (test_golden_proofs_gpu.py) requires the device prover to emit exactly
these bytes."""
import importlib.util
import os

import pytest

from conftest import load_golden

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
    fresh = lambda: mj.transcript.StandardTranscript(c, b"PlonkProof")
    assert V.verify(pc, fresh(), vk, [], proof, None, None, open_key=V.open_key_for_testing(pc, srs_beta))
    ch = V.compute_challenges(fresh(), vk, [], V.deserialize_proof(pc, proof))
    assert {name: "%x" % ch[name] for name in vec["challenges"]} == vec["challenges"]
    bad = bytearray(proof)
    bad[-40 if vec["plookup_comms"] is None else -2] ^= 1
    assert not V.verify(pc, fresh(), vk, [], bytes(bad), pyref.g1_gen(pc), srs_beta)


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
    fresh = lambda: mj.transcript.StandardTranscript(c, b"PlonkLinkingProof")
    layout = L.GroupLayout(*vec["layout"])
    open_key = V.open_key_for_testing(pc, srs_beta)
    assert L.verify_link_proof(pc, fresh(), wire0[0], wire0[1], q, o, layout, None, open_key=open_key)
    assert L.quotient_challenge(fresh(), wire0[0], wire0[1], q) == int(vec["eta"], 16)
    al, off, size = vec["layout"]
    assert not L.verify_link_proof(pc, fresh(), wire0[0], wire0[1], q, o, L.GroupLayout(al, off, size + 1), srs_beta)
    assert not L.verify_link_proof(pc, fresh(), wire0[1], wire0[0], q, o, layout, srs_beta)
