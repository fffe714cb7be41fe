"""tests/golden/make_proof_golden.py -- generates tests/golden/proof_vectors.json on the CPU.

Whole proofs of the reference's bench circuit (plonk/benches/bench.rs:29-46) by the big-int restatements alone
(oracle/pyref_circuit.py builds the circuit, oracle/pyref_snark.py proves it with schoolbook polynomial arithmetic), with the
reference's deterministic randomness: `test_rng` (ChaCha12, fixed seed) draws the SRS trapdoor first and then the blinders in
the prover's order.  Everything used here lives under oracle/ -- circuit, prover, ChaCha `test_rng` / `Fr::rand` /
`compute_coset_representatives` (oracle/pyref_rng.py), Merlin transcript and ark-serialize encoders (oracle/pyref_fs.py), each
written from the crates' published definitions: NOTHING is imported from the product package, so "the device prover emits
these bytes" (tests/test_golden_proofs_gpu.py) checks the product's transcript, rng and serialisation against a second text.
The reference holds no proof vector and cannot be built here, so these are restatement vectors (integration/rust/ holds the
Rust program that prints the reference's own bytes for the same inputs; tests/golden/ref_*.json, when present, are compared too).

    python tests/golden/make_proof_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as P  # noqa: E402
import pyref_circuit as PC  # noqa: E402
import pyref_snark as PS  # noqa: E402
import pyref_fs as FS  # noqa: E402
import pyref_rng as RNG  # noqa: E402

CASES = [(0, "TurboPlonk", 20, 8), (1, "TurboPlonk", 20, 8), (1, "UltraPlonk", 20, 3), (0, "UltraPlonk", 24, 4)]


def build(curve_id, plonk_type, num_gates, range_bits, rng=None, want_core=False, reference_setup=False):
    """reference_setup: the SRS of `universal_setup_for_testing` (plonk/src/proof_system/snark.rs:495-517) -- beta = Fr::rand, then
    g = G1::rand and h = G2::rand from the same `test_rng`, powers_of_g[i] = beta^i g -- instead of the curve's standard generator:
    what plonk/benches/bench.rs and the reference's own tests prove over.  oracle/pyref_rng.py restates the two samplers."""
    pc = P.CURVES[curve_id]
    ultra = plonk_type == "UltraPlonk"
    W = 6 if ultra else 5
    # k depends on the domain size: build once to learn n, then with the real representatives
    n = PC.bench_circuit(pc, num_gates, ultra, range_bits, list(range(1, W + 1)))[0]
    k = RNG.compute_coset_representatives(pc, W, n)
    n, wires, witness, sel, sigma, tables = PC.bench_circuit(pc, num_gates, ultra, range_bits, k)
    srs_g = None
    if reference_setup:
        assert rng is None
        rng = RNG.test_rng()
        srs_beta, srs_g = RNG.universal_setup_for_testing(pc, rng)
    elif rng is None:
        rng = RNG.test_rng()
        srs_beta = RNG.fr_rand(pc, rng)
    else:                                                                 # a later proof on the same stream: the trapdoor was its first draw
        first = RNG.test_rng()
        srs_beta = RNG.fr_rand(pc, first)
    blind = RNG.draw_blinders(pc, rng, W, ultra)
    w_vals = [[witness[v] for v in wires[i]] for i in range(W)]
    g1 = lambda p: FS.g1_bytes(pc, p)
    fr = lambda x: FS.fr_bytes(pc, x)
    out = PS.prove(pc, n.bit_length() - 1, sel, sigma, k, w_vals, [0] * n, [], blind, srs_beta, FS.StandardTranscript(pc, b"PlonkProof"),
                   g1, fr, plookup=tables, srs_g=srs_g)
    vk = out["vk"]
    rec = {"curve": curve_id, "plonk_type": plonk_type, "num_gates": num_gates, "range_bit_len": range_bits, "domain_size": n,
           "srs_beta": "%x" % srs_beta, "k": ["%x" % x for x in k],
           "selector_comms": [g1(p).hex() for p in vk["selector_comms"]], "sigma_comms": [g1(p).hex() for p in vk["sigma_comms"]],
           "plookup_comms": None, "challenges": {name: "%x" % v for name, v in out["challenges"].items()}, "proof": out["proof"].hex()}
    if ultra:
        rec["plookup_comms"] = {name: g1(p).hex() for name, p in vk["plookup"].items()}
    if reference_setup:
        rec["setup"] = "universal_setup_for_testing"
        rec["srs_g"] = ["%x" % srs_g[0], "%x" % srs_g[1]]
    return (rec, out) if want_core else rec


# general circuits (oracle/pyref_circuit.py general_circuit / general_ultra_circuit): (curve, plonk type, log2 domain size, builder seed)
GENERAL_CASES = [(0, "TurboPlonk", 4, 11), (1, "TurboPlonk", 5, 12), (1, "UltraPlonk", 4, 13), (0, "UltraPlonk", 5, 14)]


def general_instance(curve_id, plonk_type, log_n, seed):
    """The circuit of one general case: (selector values, sigma values, k, wire values, public-input vector, public input, tables | None)."""
    import random
    pc = P.CURVES[curve_id]
    rnd = random.Random(seed)
    tables = None
    if plonk_type == "UltraPlonk":
        sel, sigma, k, w, pi, tables = PC.general_ultra_circuit(pc, log_n, rnd)
    else:
        sel, sigma, k, w, pi = PC.general_circuit(pc, log_n, rnd)
    pub = pi[:4]                                                          # the public input sits on rows 0 .. 3 (row 3 is non-zero)
    assert pub[3] != 0 and not any(pi[4:])
    return sel, sigma, k, w, pi, pub, tables


def build_general(curve_id, plonk_type, log_n, seed):
    """A whole proof of a GENERAL circuit -- a non-zero public input, addition / multiplication / x^5 gates, copy constraints over all
    wires, key and range lookups (UltraPlonk) -- by the schoolbook prover, `test_rng` drawing the SRS trapdoor and then the blinders as in
    `build`.  The same instance is rebuilt from (log_n, seed) by the GPU tests, which must emit these bytes from the Python mirror, the
    round-level C ABI and `mzk_prove file`."""
    pc = P.CURVES[curve_id]
    ultra = plonk_type == "UltraPlonk"
    W = 6 if ultra else 5
    sel, sigma, k, w, pi, pub, tables = general_instance(curve_id, plonk_type, log_n, seed)
    rng = RNG.test_rng()
    srs_beta = RNG.fr_rand(pc, rng)
    blind = RNG.draw_blinders(pc, rng, W, ultra)
    g1 = lambda p: FS.g1_bytes(pc, p)
    out = PS.prove(pc, log_n, sel, sigma, k, w, pi, pub, blind, srs_beta, FS.StandardTranscript(pc, b"PlonkProof"), g1, lambda x: FS.fr_bytes(pc, x),
                   plookup=tables)
    vk = out["vk"]
    rec = {"curve": curve_id, "plonk_type": plonk_type, "log_n": log_n, "seed": seed, "domain_size": 1 << log_n, "srs_beta": "%x" % srs_beta,
           "k": ["%x" % x for x in k], "public_input": ["%x" % x for x in pub],
           "gates": {"addition": sum(1 for j in range(1 << log_n) if sel[0][j] and sel[10][j]), "multiplication": sum(1 for v in sel[4] if v),
                     "x^5": sum(1 for v in sel[6] if v), "constant": sum(1 for v in sel[11] if v), "lookup": sum(1 for v in sel[13] if v) if ultra else 0},
           "selector_comms": [g1(p).hex() for p in vk["selector_comms"]], "sigma_comms": [g1(p).hex() for p in vk["sigma_comms"]],
           "plookup_comms": {name: g1(p).hex() for name, p in vk["plookup"].items()} if ultra else None,
           "challenges": {name: "%x" % v for name, v in out["challenges"].items()}, "proof": out["proof"].hex()}
    return rec


BATCH_CASES = [(0, "TurboPlonk", (25, 28, 31), 8), (1, "UltraPlonk", (36, 40), 4)]


def build_batch(curve_id, plonk_type, gates, range_bits):
    """PlonkKzgSnark::batch_prove over bench circuits of one domain size, `test_rng` draws in batch_prove_internal's order
    (oracle/pyref_rng.py::draw_batch_blinders), by the restatements (oracle/pyref_snark.py::batch_prove)."""
    pc = P.CURVES[curve_id]
    ultra = plonk_type == "UltraPlonk"
    W = 6 if ultra else 5
    n = PC.bench_circuit(pc, gates[0], ultra, range_bits, list(range(1, W + 1)))[0]
    k = RNG.compute_coset_representatives(pc, W, n)
    rng = RNG.test_rng()
    srs_beta = RNG.fr_rand(pc, rng)
    blinds, quot = RNG.draw_batch_blinders(pc, rng, W, [ultra] * len(gates))
    instances = []
    for g, bl in zip(gates, blinds):
        n_g, wires, witness, sel, sigma, tables = PC.bench_circuit(pc, g, ultra, range_bits, k)
        assert n_g == n
        instances.append({"selector_vals": sel, "sigma_vals": sigma, "k": k, "wire_vals": [[witness[v] for v in wires[i]] for i in range(W)],
                          "pi_vals": [0] * n, "blind": bl, "plookup": tables})
    g1 = lambda p: FS.g1_bytes(pc, p)
    out = PS.batch_prove(pc, n.bit_length() - 1, instances, [[] for _ in gates], quot, srs_beta, FS.StandardTranscript(pc, b"PlonkProof"),
                         g1, lambda x: FS.fr_bytes(pc, x))
    return {"curve": curve_id, "plonk_type": plonk_type, "gates": list(gates), "range_bit_len": range_bits, "domain_size": n, "srs_beta": "%x" % srs_beta,
            "challenges": {name: "%x" % v for name, v in out["challenges"].items()}, "batch_proof": out["proof"].hex()}


LINK_CASES = [(1, (20, 22), (4, 3, 9)), (0, (40, 30), (5, 6, 14))]


def build_link(curve_id, gates, layout_args):
    """prove_with_link_hint on two TurboPlonk bench circuits of one domain size (consecutive draws of one `test_rng`), then
    PlonkKzgSnark::link_proofs over the given GroupLayout (wire 0 of the bench circuit holds the running sums 0, 1, 2, ..:
    rows below both gate counts carry the same witnesses), all by the restatements (oracle/pyref_linking.py)."""
    import pyref_linking as L
    pc = P.CURVES[curve_id]
    rng = RNG.test_rng()
    srs_beta = RNG.fr_rand(pc, rng)
    recs, cores = zip(*[build(curve_id, "TurboPlonk", g, 8, rng=rng, want_core=True) for g in gates])
    assert recs[0]["domain_size"] == recs[1]["domain_size"]
    G = P.g1_gen(pc)
    a = [core["core"]["wire_polys"][0] for core in cores]
    comms = [P.g1_mul(pc, core["core"]["commit_dlogs"]["wires"][0], G) for core in cores]
    lp = L.link_proofs(pc, a[0], a[1], comms[0], comms[1], L.GroupLayout(*layout_args), srs_beta, FS.StandardTranscript(pc, b"PlonkLinkingProof"))
    g1 = lambda p: FS.g1_bytes(pc, p)
    return {"curve": curve_id, "gates": list(gates), "layout": list(layout_args), "srs_beta": "%x" % srs_beta, "proofs": [r["proof"] for r in recs],
            "eta": "%x" % lp["eta"], "link_proof": L.serialize_link_proof(g1, lp["quotient_commitment"], lp["opening_proof"]).hex()}


if __name__ == "__main__":
    vectors = [build(*case) for case in CASES]
    with open(os.path.join(HERE, "proof_vectors.json"), "w") as f:
        json.dump(vectors, f, indent=1)
    print("wrote", len(vectors), "proof vectors:", [len(v["proof"]) // 2 for v in vectors], "bytes")
    refsetup = [build(*case, reference_setup=True) for case in CASES]
    with open(os.path.join(HERE, "proof_vectors_refsetup.json"), "w") as f:
        json.dump(refsetup, f, indent=1)
    print("wrote", len(refsetup), "proof vectors over universal_setup_for_testing's SRS (g = G1::rand)")
    links = [build_link(*case) for case in LINK_CASES]
    with open(os.path.join(HERE, "link_vectors.json"), "w") as f:
        json.dump(links, f, indent=1)
    print("wrote", len(links), "link vectors")
    general = [build_general(*case) for case in GENERAL_CASES]
    with open(os.path.join(HERE, "general_proof_vectors.json"), "w") as f:
        json.dump(general, f, indent=1)
    print("wrote", len(general), "general-circuit proof vectors:", [(v["plonk_type"], v["domain_size"], v["gates"]) for v in general])
    batches = [build_batch(*case) for case in BATCH_CASES]
    with open(os.path.join(HERE, "batch_vectors.json"), "w") as f:
        json.dump(batches, f, indent=1)
    print("wrote", len(batches), "batch vectors:", [len(v["batch_proof"]) // 2 for v in batches], "bytes")
