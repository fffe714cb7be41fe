"""tests/golden/make_proof_golden.py -- generates tests/golden/proof_vectors.json on the CPU.

Whole proofs of the reference's bench circuit (plonk/benches/bench.rs:29-46) by the big-int restatements alone
(oracle/pyref_circuit.py builds the circuit, oracle/pyref_snark.py proves it with schoolbook polynomial arithmetic), with the
reference's deterministic randomness: `test_rng` (ChaCha12, zero seed) draws the SRS trapdoor first and then the blinders in
the prover's order (mpc-jellyfish_amd/rng.py, pure Python, pinned by the ChaCha KATs of tests/test_transcript.py).
The reference holds no proof vector and cannot be built here, so these are restatement vectors: the CPU suite checks that the
restated verifier accepts them (pairing form), the GPU suite that the device prover emits exactly these bytes.

    python tests/golden/make_proof_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import pyref as P  # noqa: E402
import pyref_circuit as PC  # noqa: E402
import pyref_snark as PS  # noqa: E402
import mpc_jellyfish_amd as mj  # noqa: E402  (params / rng / transcript: pure Python, no GPU)

CASES = [(0, "TurboPlonk", 20, 8), (1, "TurboPlonk", 20, 8), (1, "UltraPlonk", 20, 3), (0, "UltraPlonk", 24, 4)]


def build(curve_id, plonk_type, num_gates, range_bits, rng=None, want_core=False):
    c, pc = mj.params.CURVES[curve_id], P.CURVES[curve_id]
    ultra = plonk_type == "UltraPlonk"
    W = 6 if ultra else 5
    # k depends on the domain size: build once to learn n, then with the real representatives
    n = PC.bench_circuit(pc, num_gates, ultra, range_bits, list(range(1, W + 1)))[0]
    k = mj.rng.compute_coset_representatives(c, W, n)
    n, wires, witness, sel, sigma, tables = PC.bench_circuit(pc, num_gates, ultra, range_bits, k)
    if rng is None:
        rng = mj.rng.test_rng()
        srs_beta = mj.rng.fr_rand(c, rng)
    else:                                                                 # a later proof on the same stream: the trapdoor was its first draw
        first = mj.rng.test_rng()
        srs_beta = mj.rng.fr_rand(c, first)
    bl = mj.snark.draw_blinders(c, rng, W, ultra)
    blind = {"wires": bl.wires, "z": bl.z, "quot": bl.quot, "h": bl.h, "prod_lookup": bl.prod_lookup}
    w_vals = [[witness[v] for v in wires[i]] for i in range(W)]
    g1 = lambda p: mj.transcript.g1_bytes(c, p)
    fr = lambda x: mj.transcript.fr_bytes(c, x)
    out = PS.prove(pc, n.bit_length() - 1, sel, sigma, k, w_vals, [0] * n, [], blind, srs_beta, mj.transcript.StandardTranscript(c, b"PlonkProof"),
                   g1, fr, plookup=tables)
    vk = out["vk"]
    rec = {"curve": curve_id, "plonk_type": plonk_type, "num_gates": num_gates, "range_bit_len": range_bits, "domain_size": n,
           "srs_beta": "%x" % srs_beta, "k": ["%x" % x for x in k],
           "selector_comms": [g1(p).hex() for p in vk["selector_comms"]], "sigma_comms": [g1(p).hex() for p in vk["sigma_comms"]],
           "plookup_comms": None, "challenges": {name: "%x" % v for name, v in out["challenges"].items()}, "proof": out["proof"].hex()}
    if ultra:
        rec["plookup_comms"] = {name: g1(p).hex() for name, p in vk["plookup"].items()}
    return (rec, out) if want_core else rec


BATCH_CASES = [(0, "TurboPlonk", (25, 28, 31), 8), (1, "UltraPlonk", (36, 40), 4)]


def build_batch(curve_id, plonk_type, gates, range_bits):
    """PlonkKzgSnark::batch_prove over bench circuits of one domain size, `test_rng` draws in batch_prove_internal's order
    (snark.draw_batch_blinders), by the restatements (oracle/pyref_snark.py::batch_prove)."""
    c, pc = mj.params.CURVES[curve_id], P.CURVES[curve_id]
    ultra = plonk_type == "UltraPlonk"
    W = 6 if ultra else 5
    n = PC.bench_circuit(pc, gates[0], ultra, range_bits, list(range(1, W + 1)))[0]
    k = mj.rng.compute_coset_representatives(c, W, n)
    rng = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng)
    blinds, quot = mj.snark.draw_batch_blinders(c, rng, W, [ultra] * len(gates))
    instances = []
    for g, bl in zip(gates, blinds):
        n_g, wires, witness, sel, sigma, tables = PC.bench_circuit(pc, g, ultra, range_bits, k)
        assert n_g == n
        instances.append({"selector_vals": sel, "sigma_vals": sigma, "k": k, "wire_vals": [[witness[v] for v in wires[i]] for i in range(W)],
                          "pi_vals": [0] * n, "blind": {"wires": bl.wires, "z": bl.z, "h": bl.h, "prod_lookup": bl.prod_lookup}, "plookup": tables})
    g1 = lambda p: mj.transcript.g1_bytes(c, p)
    out = PS.batch_prove(pc, n.bit_length() - 1, instances, [[] for _ in gates], quot, srs_beta, mj.transcript.StandardTranscript(c, b"PlonkProof"),
                         g1, lambda x: mj.transcript.fr_bytes(c, x))
    return {"curve": curve_id, "plonk_type": plonk_type, "gates": list(gates), "range_bit_len": range_bits, "domain_size": n, "srs_beta": "%x" % srs_beta,
            "challenges": {name: "%x" % v for name, v in out["challenges"].items()}, "batch_proof": out["proof"].hex()}


LINK_CASES = [(1, (20, 22), (4, 3, 9)), (0, (40, 30), (5, 6, 14))]


def build_link(curve_id, gates, layout_args):
    """prove_with_link_hint on two TurboPlonk bench circuits of one domain size (consecutive draws of one `test_rng`), then
    PlonkKzgSnark::link_proofs over the given GroupLayout (wire 0 of the bench circuit holds the running sums 0, 1, 2, ..:
    rows below both gate counts carry the same witnesses), all by the restatements (oracle/pyref_linking.py)."""
    import pyref_linking as L
    c, pc = mj.params.CURVES[curve_id], P.CURVES[curve_id]
    rng = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng)
    recs, cores = zip(*[build(curve_id, "TurboPlonk", g, 8, rng=rng, want_core=True) for g in gates])
    assert recs[0]["domain_size"] == recs[1]["domain_size"]
    G = P.g1_gen(pc)
    a = [core["core"]["wire_polys"][0] for core in cores]
    comms = [P.g1_mul(pc, core["core"]["commit_dlogs"]["wires"][0], G) for core in cores]
    lp = L.link_proofs(pc, a[0], a[1], comms[0], comms[1], L.GroupLayout(*layout_args), srs_beta, mj.transcript.StandardTranscript(c, b"PlonkLinkingProof"))
    g1 = lambda p: mj.transcript.g1_bytes(c, p)
    return {"curve": curve_id, "gates": list(gates), "layout": list(layout_args), "srs_beta": "%x" % srs_beta, "proofs": [r["proof"] for r in recs],
            "eta": "%x" % lp["eta"], "link_proof": L.serialize_link_proof(g1, lp["quotient_commitment"], lp["opening_proof"]).hex()}


if __name__ == "__main__":
    vectors = [build(*case) for case in CASES]
    with open(os.path.join(HERE, "proof_vectors.json"), "w") as f:
        json.dump(vectors, f, indent=1)
    print("wrote", len(vectors), "proof vectors:", [len(v["proof"]) // 2 for v in vectors], "bytes")
    links = [build_link(*case) for case in LINK_CASES]
    with open(os.path.join(HERE, "link_vectors.json"), "w") as f:
        json.dump(links, f, indent=1)
    print("wrote", len(links), "link vectors")
    batches = [build_batch(*case) for case in BATCH_CASES]
    with open(os.path.join(HERE, "batch_vectors.json"), "w") as f:
        json.dump(batches, f, indent=1)
    print("wrote", len(batches), "batch vectors:", [len(v["batch_proof"]) // 2 for v in batches], "bytes")
