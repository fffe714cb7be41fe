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


@pytest.mark.parametrize("index", [0, 1, 2, 3])
def test_golden_general_circuit_proof_vectors(pyref, index):
    """tests/golden/general_proof_vectors.json: whole proofs of GENERAL circuits (non-zero public input, addition / multiplication / x^5
    gates, copy constraints, key + range lookups) by the schoolbook prover -- regenerated identically; the restated reference verifier
    accepts them with their public input in the pairing form, and rejects a changed public input or proof byte."""
    import pyref_verifier as V
    vec = load_golden("general_proof_vectors")[index]
    gen = _generator()
    assert (vec["curve"], vec["plonk_type"], vec["log_n"], vec["seed"]) == gen.GENERAL_CASES[index]
    assert gen.build_general(*gen.GENERAL_CASES[index]) == vec, "tests/golden/general_proof_vectors.json is stale: run tests/golden/make_proof_golden.py"
    g = vec["gates"]
    assert g["addition"] and g["multiplication"] and g["x^5"] and g["constant"] and (g["lookup"] > 0) == (vec["plonk_type"] == "UltraPlonk")
    pc = pyref.CURVES[vec["curve"]]
    pub = [int(x, 16) for x in vec["public_input"]]
    assert any(pub)
    vk = golden_vk(V, pc, vec)
    vk["num_inputs"] = len(pub)
    proof = bytes.fromhex(vec["proof"])
    srs_beta = int(vec["srs_beta"], 16)
    fresh = lambda: FS.StandardTranscript(pc, b"PlonkProof")
    assert V.verify(pc, fresh(), vk, pub, proof, None, None, open_key=V.open_key_for_testing(pc, srs_beta))
    ch = V.compute_challenges(fresh(), vk, pub, V.deserialize_proof(pc, proof))
    assert {name: "%x" % ch[name] for name in vec["challenges"]} == vec["challenges"]
    wrong = pub[:3] + [(pub[3] + 1) % pc.r]
    assert not V.verify(pc, fresh(), vk, wrong, proof, pyref.g1_gen(pc), srs_beta)
    bad = bytearray(proof)
    bad[-40 if vec["plookup_comms"] is None else -2] ^= 1
    assert not V.verify(pc, fresh(), vk, pub, bytes(bad), pyref.g1_gen(pc), srs_beta)


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


@pytest.mark.parametrize("name", ["proof_vectors", "proof_vectors_refsetup", "general_proof_vectors"])
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
        for f in ("curve", "plonk_type", "domain_size", "srs_beta") + (("srs_g",) if "srs_g" in a else ()) + (("num_gates",) if "num_gates" in a else ("log_n", "seed", "public_input")):
            assert a[f] == b[f]
        assert [int(x, 16) for x in a["k"]] == [int(x, 16) for x in b["k"]]
        assert a["selector_comms"] == b["selector_comms"] and a["sigma_comms"] == b["sigma_comms"]
        assert a["proof"] == b["proof"], "proof bytes differ from the reference's"


# ---- the reference-made families of integration/rust/src/bin/gen_fixtures.rs (round 5): batch, link, general circuits ----------------
@pytest.mark.parametrize("name,fields", [("batch_vectors", ("batch_proof",)), ("link_vectors", ("proofs", "link_proof"))])
def test_reference_batch_and_link_fixtures_when_present(name, fields):
    """tests/golden/ref_batch_vectors.json / ref_link_vectors.json = `PlonkKzgSnark::batch_prove` / `prove_with_link_hint` + `link_proofs` of
    the reference itself on the inputs of the committed vectors: the restatements' bytes must equal them."""
    import json
    path = os.path.join(HERE, "golden", "ref_%s.json" % name)
    if not os.path.exists(path):
        pytest.skip("reference fixtures absent (integration/rust has not been run): parity unpinned")
    ref, ours = json.load(open(path)), load_golden(name)
    assert len(ref) == len(ours)
    for a, b in zip(ours, ref):
        assert a["curve"] == b["curve"] and a["gates"] == b["gates"] and a["srs_beta"] == b["srs_beta"]
        for f in fields:
            assert a[f] == b[f], "%s differs from the reference's" % f


def oracle_proof_of_circuit_file(pyref, mj, blob):
    """The schoolbook prover (oracle/pyref_snark.py) on a circuit handed over in the circuit-file format: `test_rng` draws the SRS trapdoor,
    then the blinders.  Returns (proof bytes, public input, srs_beta)."""
    from importlib import import_module
    import gc
    from conftest import fr_from_mont_limbs
    import pyref_snark as PS
    io = import_module("mpc-jellyfish_amd.circuit_io")
    cf = io.read_circuit(blob)
    pc = pyref.CURVES[cf["curve_id"]]
    ints = lambda a: fr_from_mont_limbs(pc, a)
    n, W = 1 << cf["log_n"], cf["num_wire_types"]
    ultra = W == 6
    pub = ints(cf["pub_values"])
    pi = [0] * n
    for row, v in zip(cf["pub_rows"], pub):
        pi[row] = v
    tables = {name: ints(cf["tables"][name]) for name in io.TABLES} if ultra else None
    rng = RNG.test_rng()
    srs_beta = RNG.fr_rand(pc, rng)
    blind = RNG.draw_blinders(pc, rng, W, ultra)
    out = PS.prove(pc, cf["log_n"], [ints(s) for s in cf["selectors"]], [ints(s) for s in cf["sigmas"]], ints(cf["k"]), [ints(w) for w in cf["wires"]],
                   pi, pub, blind, srs_beta, FS.StandardTranscript(pc, b"PlonkProof"), lambda p: FS.g1_bytes(pc, p), lambda x: FS.fr_bytes(pc, x), plookup=tables)
    gc.collect()
    return out["proof"], pub, srs_beta, out["vk"]


@pytest.mark.parametrize("index", [0, 2])
def test_oracle_proves_a_circuit_file(pyref, mj, tmp_path, index):
    """The path the reference-made general circuits take (below), exercised on the committed general vectors: write the instance in the
    circuit-file format, read it back, prove it with the schoolbook oracle -> the committed bytes."""
    from importlib import import_module
    io = import_module("mpc-jellyfish_amd.circuit_io")
    vec = load_golden("general_proof_vectors")[index]
    gen = _generator()
    sel, sigma, k, w, pi, pub, tables = gen.general_instance(*gen.GENERAL_CASES[index])
    path = str(tmp_path / "c.bin")
    io.write_circuit(path, vec["curve"], vec["log_n"], sel, sigma, k, w, pub_input=pub, tables=tables)
    proof, pub2, beta, _ = oracle_proof_of_circuit_file(pyref, mj, open(path, "rb").read())
    assert proof.hex() == vec["proof"] and pub2 == pub and "%x" % beta == vec["srs_beta"]
    with pytest.raises(ValueError):
        io.read_circuit(open(path, "rb").read()[:-8])


def test_reference_general_circuits_when_present(pyref, mj):
    """tests/golden/ref_general_circuits.json: GENERAL circuits built through the reference's own gadgets (public inputs, add / mul / pow5 /
    lc gates, shared variables, range + key lookups), exported as circuit files and proved by the reference (gen_fixtures.rs `general`).
    The schoolbook oracle on the same file must emit the reference's proof and verifying-key bytes (domains up to 2^7: larger cases are
    left to the GPU test, tests/test_golden_proofs_gpu.py)."""
    import json
    path = os.path.join(HERE, "golden", "ref_general_circuits.json")
    if not os.path.exists(path):
        pytest.skip("reference fixtures absent (integration/rust has not been run): parity unpinned")
    for case in json.load(open(path)):
        if case["domain_size"] > 128:
            continue
        pc = pyref.CURVES[case["curve"]]
        proof, pub, beta, vk = oracle_proof_of_circuit_file(pyref, mj, bytes.fromhex(case["circuit_file"]))
        assert "%x" % beta == case["srs_beta"] and ["%x" % x for x in pub] == case["public_input"]
        assert [FS.g1_bytes(pc, p).hex() for p in vk["selector_comms"]] == case["selector_comms"]
        assert [FS.g1_bytes(pc, p).hex() for p in vk["sigma_comms"]] == case["sigma_comms"]
        assert proof.hex() == case["proof"], "proof bytes differ from the reference's"
