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
    import mirror_prover as MP
    for mode in ("from the SRS", "from beta", "off"):
        from_beta = mj.UnivariateProverParam.gen_lagrange_srs_for_testing(c, srs_beta, cs.n, g=g) if mode == "from beta" else None
        # the product (round-level C ABI) and the test-side sequencing of the primitives, both with the same key
        pk = mj.snark.preprocess(ck, cs, lagrange=mode == "from the SRS", lagrange_ck=from_beta)           # (lagrange=None, the default: from 2^13 gates on)
        assert (pk.lagrange_ck is not None) == (mode != "off")
        mk = MP.preprocess(ck, cs, lagrange=False)
        mk.lagrange_ck = pk.lagrange_ck
        for name, prove, key in (("product", mj.snark.prove, pk), ("mirror", MP.prove, mk)):
            _, proof_bytes = prove(_rng_after_setup(mj, c, "srs_g" in vec), cs, key)
            assert proof_bytes.hex() == vec["proof"], (mode, name)
        mk.lagrange_ck = None
        mk.release()
        pk.release()
        if from_beta is not None:
            from_beta.release()
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


# ---- general circuits: oracle-made bytes for a non-zero public input, every arithmetic gate family, copy constraints, lookups --------
def _general_case(mj, pyref, vec):
    """The instance of one general golden vector, rebuilt from (log_n, seed) by the oracle's builder, in the product's forms."""
    import random
    import numpy as np
    import pyref_circuit as PC
    from conftest import fr_mont_limbs
    c, pc = mj.params.CURVES[vec["curve"]], pyref.CURVES[vec["curve"]]
    ultra = vec["plonk_type"] == "UltraPlonk"
    rnd = random.Random(vec["seed"])
    tabs = None
    if ultra:
        sel, sig, k, w, pi, tabs = PC.general_ultra_circuit(pc, vec["log_n"], rnd)
    else:
        sel, sig, k, w, pi = PC.general_circuit(pc, vec["log_n"], rnd)
    assert ["%x" % x for x in pi[:4]] == vec["public_input"] and ["%x" % x for x in k] == vec["k"]
    dom = mj.Radix2EvaluationDomain(c, vec["log_n"])
    kw = {"plookup": {name: dom.ifft(fr_mont_limbs(c, tabs[key])) for name, key in
                      zip(mj.plonk.PLOOKUP_TABLE_POLYS, ("range", "key", "table_dom_sep", "q_dom_sep"))}} if ultra else {}
    sel_p, sig_p = [dom.ifft(fr_mont_limbs(c, s)) for s in sel], [dom.ifft(fr_mont_limbs(c, s)) for s in sig]
    return c, ultra, (sel, sig, k, w, pi, tabs), sel_p, sig_p, np.stack([fr_mont_limbs(c, col) for col in w]), kw


@pytest.mark.parametrize("index", [0, 1, 2, 3])
def test_general_circuit_golden_proofs_from_all_three_hosts(gpu, mj, pyref, tmp_path, index):
    """tests/golden/general_proof_vectors.json (oracle/pyref_snark.py on oracle/pyref_circuit.py's general circuits: non-zero public
    input, add / mul / x^5 gates, copy constraints over all wires, key + range lookups): the SAME bytes -- verifying key and proof --
    from (i) the Python mirror (primitive-level sequencing), (ii) the round-level C ABI (mzk_prover_*, through ctypes) and (iii) the
    compiled host reading the circuit from a file (`mzk_prove <curve> file`)."""
    import json
    import os
    import subprocess
    from importlib import import_module
    from conftest import fr_mont_limbs
    vec = load_golden("general_proof_vectors")[index]
    c, ultra, raw, sel_p, sig_p, wires, kw = _general_case(mj, pyref, vec)
    sel, sig, k, w, pi, tabs = raw
    n, W = vec["domain_size"], 6 if ultra else 5
    pub = pi[:4]
    g1 = lambda x: mj.snark._g1(c, x).hex()

    def rng_and_key():
        rng = mj.rng.test_rng()
        beta = mj.rng.fr_rand(c, rng)
        assert "%x" % beta == vec["srs_beta"]
        return rng, beta

    rng, beta = rng_and_key()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, beta, n + 2)
    # (i) the Python mirror
    import mirror_prover as MP
    mirror = MP.TurboPlonkProver(c, n, sel_p, sig_p, k, ck, **kw)
    sel_c, sig_c = mirror.vk_commitments()
    assert [g1(x) for x in sel_c] == vec["selector_comms"] and [g1(x) for x in sig_c] == vec["sigma_comms"]
    if ultra:
        names = ("range_table_comm", "key_table_comm", "table_dom_sep_comm", "q_dom_sep_comm")
        assert dict(zip(names, [g1(x) for x in mirror.plookup_vk_commitments()])) == vec["plookup_comms"]
    blind = mj.snark.draw_blinders(c, rng, W, ultra)
    src = mj.prover.TranscriptChallenges(mirror, pub)
    core = mirror.prove(wires, fr_mont_limbs(c, pi), src, blind)
    assert mj.snark.serialize_proof(c, core).hex() == vec["proof"], "Python mirror"
    assert {name: "%x" % v for name, v in src.challenges.items()} == vec["challenges"]
    mirror.release()
    # (ii) the round-level C ABI
    native = mj.prover.TurboPlonkProver(c, n, sel_p, sig_p, k, ck, **kw)
    sel_c, sig_c = native.vk_commitments()
    assert [g1(x) for x in sel_c] == vec["selector_comms"] and [g1(x) for x in sig_c] == vec["sigma_comms"]
    rng, _ = rng_and_key()
    blind = mj.snark.draw_blinders(c, rng, W, ultra)
    core = native.prove(wires, pub, mj.prover.TranscriptChallenges(native, pub), blind)
    assert mj.snark.serialize_proof(c, core).hex() == vec["proof"], "round-level ABI"
    native.release()
    ck.release()
    # (iii) the compiled host, circuit from a file
    io = import_module("mpc-jellyfish_amd.circuit_io")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "general.bin")
    io.write_circuit(path, c, vec["log_n"], sel, sig, k, w, pub_input=pub, tables=tabs)
    out = subprocess.run([os.path.join(root, "mpc-jellyfish_amd", "mzk_prove"), str(vec["curve"]), "file", path, "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["proof_hex"] == vec["proof"], "mzk_prove file"
    assert got["vk_hex"] == "".join(vec["selector_comms"] + vec["sigma_comms"])


def _check_circuit_file_case(mj, tmp_path, case, i):
    """`mzk_prove <curve> file` and the round-level C ABI on case["circuit_file"] must emit case["proof"] (and the verifying key)."""
    import json
    import os
    import subprocess
    from importlib import import_module
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    io = import_module("mpc-jellyfish_amd.circuit_io")
    blob = bytes.fromhex(case["circuit_file"])
    f = str(tmp_path / ("ref_%d.bin" % i))
    open(f, "wb").write(blob)
    out = subprocess.run([os.path.join(root, "mpc-jellyfish_amd", "mzk_prove"), str(case["curve"]), "file", f, "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["proof_hex"] == case["proof"], "mzk_prove file"
    assert got["vk_hex"] == "".join(case["selector_comms"] + case["sigma_comms"])
    cf = io.read_circuit(blob)
    c = mj.params.CURVES[cf["curve_id"]]
    n, W = 1 << cf["log_n"], cf["num_wire_types"]
    dom = mj.Radix2EvaluationDomain(c, cf["log_n"])
    kw = {"plookup": {name: dom.ifft(cf["tables"][key]) for name, key in zip(mj.plonk.PLOOKUP_TABLE_POLYS, io.TABLES)}} if W == 6 else {}
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), n + 2)
    native = mj.prover.TurboPlonkProver(c, n, [dom.ifft(s) for s in cf["selectors"]], [dom.ifft(s) for s in cf["sigmas"]], mj.params.fr_from_mont(c, cf["k"]), ck, **kw)
    pub = mj.params.fr_from_mont(c, cf["pub_values"])
    blind = mj.snark.draw_blinders(c, rng, W, W == 6)
    core = native.prove(cf["wires"], (cf["pub_rows"], pub), mj.prover.TranscriptChallenges(native, pub), blind)
    assert mj.snark.serialize_proof(c, core).hex() == case["proof"], "round-level ABI"
    native.release()
    ck.release()


def test_circuit_file_path_on_the_committed_general_vectors(gpu, mj, pyref, tmp_path):
    """The path the reference-made general circuits take (next test), exercised on the committed oracle-made vectors."""
    from importlib import import_module
    io = import_module("mpc-jellyfish_amd.circuit_io")
    for i, vec in enumerate(load_golden("general_proof_vectors")):
        c, ultra, (sel, sig, k, w, pi, tabs), *_ = _general_case(mj, pyref, vec)
        path = str(tmp_path / ("g%d.bin" % i))
        io.write_circuit(path, c, vec["log_n"], sel, sig, k, w, pub_input=pi[:4], tables=tabs)
        _check_circuit_file_case(mj, tmp_path, dict(vec, circuit_file=open(path, "rb").read().hex()), i)


def test_reference_general_circuits_on_the_device_when_present(gpu, mj, tmp_path):
    """tests/golden/ref_general_circuits.json (gen_fixtures.rs `general`: circuits built and proved by the REFERENCE, handed over as circuit
    files): `mzk_prove <curve> file` and the round-level C ABI on the same file must emit the reference's proof bytes."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "ref_general_circuits.json")
    if not os.path.exists(path):
        pytest.skip("reference fixtures absent (integration/rust has not been run): parity unpinned")
    for i, case in enumerate(json.load(open(path))):
        _check_circuit_file_case(mj, tmp_path, case, i)
