"""CPU: the restated reference verifier (oracle/pyref_verifier.py: verifier.rs:68-251, 340-733) must accept the proofs of the
schoolbook prover restatement (oracle/pyref_plonk.py), which shares no code with it, and reject them once anything is altered.
The GPU-side use (device proofs, from their serialized bytes, challenges recomputed from the transcript) is
tests/test_verifier_gpu.py."""
import random

import pytest

from conftest import build_circuit, build_ultra_circuit


def restated_instance(pyref, curve_id, ultra, seed):
    """A satisfied circuit, its proof by pyref_plonk, and the verifying key -- commitments through the trapdoor."""
    import pyref_plonk as PP
    c = pyref.CURVES[curve_id]
    r = c.r
    rng = random.Random(seed)
    log_n = 5 if ultra else 4
    W = 6 if ultra else 5
    plookup = None
    if ultra:
        sel, sig, k, w, pi, plookup = build_ultra_circuit(c, log_n, rng)
    else:
        sel, sig, k, w, pi = build_circuit(c, log_n, rng)
    blind = {"wires": [[rng.randrange(r) for _ in range(2)] for _ in range(W)], "z": [rng.randrange(r) for _ in range(3)],
             "quot": [rng.randrange(r) for _ in range(W - 1)], "h": [[rng.randrange(r) for _ in range(3)] for _ in range(2)],
             "prod_lookup": [rng.randrange(r) for _ in range(3)]}
    ch = {x: rng.randrange(r) for x in ("tau", "beta", "gamma", "alpha", "zeta", "v")}
    srs_beta = rng.randrange(1, r)
    out = PP.prove_core(c, log_n, sel, sig, k, w, pi, blind, ch, srs_beta, plookup=plookup)
    assert out["divisible"] and out["quot_degree_ok"]
    G = pyref.g1_gen(c)
    pt = lambda dlog: pyref.g1_mul(c, dlog % r, G) if dlog % r else None
    commit = lambda poly: pt(pyref.poly_eval(c, poly, srs_beta))
    dl = out["commit_dlogs"]
    proof = {"wires_poly_comms": [pt(d) for d in dl["wires"]], "prod_perm_poly_comm": pt(dl["z"]),
             "split_quot_poly_comms": [pt(d) for d in dl["split"]], "opening_proof": pt(dl["opening"]),
             "shifted_opening_proof": pt(dl["shifted_opening"]), "wires_evals": out["wires_evals"],
             "wire_sigma_evals": out["wire_sigma_evals"], "perm_next_eval": out["perm_next_eval"], "plookup": None}
    vk = {"domain_size": 1 << log_n, "num_inputs": 4, "k": k, "selector_comms": [commit(p) for p in out["selectors"]],
          "sigma_comms": [commit(p) for p in out["sigmas"]], "plookup": None}
    if ultra:
        proof["plookup"] = {"h_poly_comms": [pt(d) for d in dl["h"]], "prod_lookup_poly_comm": pt(dl["prod_lookup"]),
                            "evals": dict(out["plookup_evals"])}
        tab = out["table_polys"]
        vk["plookup"] = {"range_table_comm": commit(tab["range"]), "key_table_comm": commit(tab["key"]),
                         "table_dom_sep_comm": commit(tab["table_dom_sep"]), "q_dom_sep_comm": commit(tab["q_dom_sep"])}
    ch["u"] = rng.randrange(r)
    return c, vk, pi[:4], proof, ch, srs_beta


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("ultra", [False, True])
def test_restated_verifier_accepts_the_restated_prover(pyref, curve_id, ultra):
    import pyref_verifier as V
    c, vk, pub, proof, ch, srs_beta = restated_instance(pyref, curve_id, ultra, 7000 + curve_id + 2 * ultra)
    G = pyref.g1_gen(c)
    accept = lambda pr, pub_=pub, ch_=ch: V.batch_verify_opening_proof(c, G, srs_beta, V.prepare_pcs_info(c, vk, pub_, pr, ch_))
    assert pub[3] != 0, "the instance has a non-trivial public input"
    assert accept(proof)
    # soundness of the check itself: every altered part is rejected
    r = c.r
    bad = dict(proof, wires_evals=[(proof["wires_evals"][0] + 1) % r] + proof["wires_evals"][1:])
    assert not accept(bad)
    bad = dict(proof, perm_next_eval=(proof["perm_next_eval"] + 1) % r)
    assert not accept(bad)
    bad = dict(proof, split_quot_poly_comms=proof["split_quot_poly_comms"][::-1])
    assert not accept(bad)
    bad = dict(proof, opening_proof=proof["shifted_opening_proof"], shifted_opening_proof=proof["opening_proof"])
    assert not accept(bad)
    assert not accept(proof, pub_=[pub[0], pub[1], pub[2], (pub[3] + 1) % r])
    assert not accept(proof, ch_=dict(ch, alpha=(ch["alpha"] + 1) % r))
    if ultra:
        pl = proof["plookup"]
        bad = dict(proof, plookup=dict(pl, evals=dict(pl["evals"], h_2_next_eval=(pl["evals"]["h_2_next_eval"] + 1) % r)))
        assert not accept(bad)
        bad = dict(proof, plookup=dict(pl, h_poly_comms=pl["h_poly_comms"][::-1]))
        assert not accept(bad)
        with pytest.raises(V.VerifyError):
            accept(dict(proof, plookup=None))
    with pytest.raises(V.VerifyError):
        accept(proof, pub_=pub[:3])


@pytest.mark.parametrize("curve_id", [0, 1])
def test_compressed_g1_round_trip_and_malformed_encodings(pyref, mj, curve_id):
    """g1_decompress inverts the boundary's G1 encoding (ark-serialize compressed: zcash flags for BLS12-381, arkworks
    short-Weierstrass flags for BN254), both y signs and infinity; malformed encodings raise."""
    import pyref_verifier as V
    c = pyref.CURVES[curve_id]
    pc = mj.params.CURVES[curve_id]
    G = pyref.g1_gen(c)
    rng = random.Random(31 + curve_id)
    signs = set()
    for _ in range(12):
        p = pyref.g1_mul(c, rng.randrange(1, c.r), G)
        b = mj.transcript.g1_bytes(pc, p)
        assert V.g1_decompress(c, b) == p
        signs.add(p[1] > c.q - p[1])
    assert signs == {True, False}
    inf = mj.transcript.g1_bytes(pc, None)
    assert V.g1_decompress(c, inf) is None
    x = 0
    while pow((x ** 3 + c.b) % c.q, (c.q - 1) // 2, c.q) == 1 or (x ** 3 + c.b) % c.q == 0:
        x += 1                                                            # an x with no point above it
    enc = bytearray(x.to_bytes(48, "big")) if curve_id == 0 else bytearray(x.to_bytes(32, "little"))
    if curve_id == 0:
        enc[0] |= 0x80
    with pytest.raises(V.VerifyError):
        V.g1_decompress(c, bytes(enc))
    with pytest.raises(V.VerifyError):
        V.g1_decompress(c, inf[:-1])
    # a proof cut short, or with bytes appended, does not deserialize
    vec = lambda items: len(items).to_bytes(8, "little") + b"".join(items)
    g = mj.transcript.g1_bytes(pc, G)
    fr = (5).to_bytes(32, "little")
    blob = vec([g] * 5) + g + vec([g] * 5) + g + g + vec([fr] * 5) + vec([fr] * 4) + fr + b"\x00"
    pr = V.deserialize_proof(c, blob)
    assert pr["wires_poly_comms"] == [G] * 5 and pr["wire_sigma_evals"] == [5] * 4 and pr["plookup"] is None
    for bad in (blob[:-1], blob + b"\x00", blob[:-1] + b"\x02"):
        with pytest.raises(V.VerifyError):
            V.deserialize_proof(c, bad)


@pytest.mark.parametrize("curve_id,ultra", [(0, False), (1, True)])
def test_restated_batch_verifier_accepts_the_restated_batch_prover(pyref, curve_id, ultra):
    """An aggregated proof over three instances (snark.rs:201-469 restated by pyref_plonk.batch_prove_core) against
    prepare_pcs_info over several verifying keys (verifier.rs:68-184: alpha_bases, running v / uv powers)."""
    import pyref_plonk as PP
    import pyref_verifier as V
    c = pyref.CURVES[curve_id]
    r = c.r
    rng = random.Random(8100 + curve_id)
    log_n = 5 if ultra else 4
    W = 6 if ultra else 5
    instances = []
    for _ in range(3):
        plookup = None
        if ultra:
            sel, sig, k, w, pi, plookup = build_ultra_circuit(c, log_n, rng)
        else:
            sel, sig, k, w, pi = build_circuit(c, log_n, rng)
        blind = {"wires": [[rng.randrange(r) for _ in range(2)] for _ in range(W)], "z": [rng.randrange(r) for _ in range(3)],
                 "h": [[rng.randrange(r) for _ in range(3)] for _ in range(2)], "prod_lookup": [rng.randrange(r) for _ in range(3)]}
        instances.append({"selector_vals": sel, "sigma_vals": sig, "k": k, "wire_vals": w, "pi_vals": pi, "blind": blind, "plookup": plookup})
    ch = {x: rng.randrange(r) for x in ("tau", "beta", "gamma", "alpha", "zeta", "v")}
    srs_beta = rng.randrange(1, r)
    out = PP.batch_prove_core(c, log_n, instances, ch, [rng.randrange(r) for _ in range(W - 1)], srs_beta)
    assert out["divisible"] and out["quot_degree_ok"]
    G = pyref.g1_gen(c)
    pt = lambda dlog: pyref.g1_mul(c, dlog % r, G) if dlog % r else None
    commit = lambda poly: pt(pyref.poly_eval(c, poly, srs_beta))
    dl = out["commit_dlogs"]
    bp = {"wires_poly_comms_vec": [[pt(d) for d in ws] for ws in dl["wires"]], "prod_perm_poly_comms_vec": [pt(d) for d in dl["z"]],
          "poly_evals_vec": [{k_: o[k_] for k_ in ("wires_evals", "wire_sigma_evals", "perm_next_eval")} for o in out["instances"]],
          "plookup_proofs_vec": [None] * 3, "split_quot_poly_comms": [pt(d) for d in dl["split"]], "opening_proof": pt(dl["opening"]),
          "shifted_opening_proof": pt(dl["shifted_opening"])}
    vks, pubs = [], []
    for i, (inst, o) in enumerate(zip(instances, out["instances"])):
        vk = {"domain_size": 1 << log_n, "num_inputs": 4, "k": inst["k"], "selector_comms": [commit(p) for p in o["selectors"]],
              "sigma_comms": [commit(p) for p in o["sigmas"]], "plookup": None}
        if ultra:
            bp["plookup_proofs_vec"][i] = {"h_poly_comms": [pt(d) for d in dl["h"][i]], "prod_lookup_poly_comm": pt(dl["prod_lookup"][i]),
                                           "evals": dict(o["plookup_evals"])}
            tab = o["table_polys"]
            vk["plookup"] = {"range_table_comm": commit(tab["range"]), "key_table_comm": commit(tab["key"]),
                             "table_dom_sep_comm": commit(tab["table_dom_sep"]), "q_dom_sep_comm": commit(tab["q_dom_sep"])}
        vks.append(vk)
        pubs.append(inst["pi_vals"][:4])
    ch["u"] = rng.randrange(r)
    accept = lambda bp_, vks_=vks, pubs_=pubs: V.batch_verify_opening_proof(c, G, srs_beta, V.prepare_pcs_info_batch(c, vks_, pubs_, bp_, ch))
    assert accept(bp)
    # instance order matters everywhere (snark.rs test :1735-1738 swaps public inputs)
    other_pub = [pubs[0][:3] + [(pubs[0][3] + 1) % r]] + pubs[1:]
    assert not accept(bp, pubs_=other_pub)
    assert not accept(bp, vks_=[vks[1], vks[0], vks[2]])
    assert not accept(dict(bp, prod_perm_poly_comms_vec=bp["prod_perm_poly_comms_vec"][::-1]))
    ev2 = dict(bp["poly_evals_vec"][2], perm_next_eval=(bp["poly_evals_vec"][2]["perm_next_eval"] + 1) % r)
    assert not accept(dict(bp, poly_evals_vec=bp["poly_evals_vec"][:2] + [ev2]))
    with pytest.raises(V.VerifyError):
        accept(bp, vks_=vks[:2])
    # one instance of it alone is the plain single-instance case: the aggregate of one equals Proof -> BatchProof
    assert V.batch_proof_from({"wires_poly_comms": 1, "prod_perm_poly_comm": 2, "wires_evals": 3, "wire_sigma_evals": 4, "perm_next_eval": 5,
                               "plookup": None, "split_quot_poly_comms": 6, "opening_proof": 7, "shifted_opening_proof": 8})["poly_evals_vec"] == \
        [{"wires_evals": 3, "wire_sigma_evals": 4, "perm_next_eval": 5}]


@pytest.mark.parametrize("curve_id", [0, 1])
def test_pairing_restatement_and_the_reference_final_check(pyref, curve_id):
    """oracle/pyref_pairing.py checks itself (generators, twist, bilinearity, non-degeneracy); the verifier's final step
    evaluated as the reference evaluates it -- multi_pairing([A, -B], [beta_h, h]) == 1 (verifier.rs:226-250) -- agrees with
    its trapdoor form on an accepted and on a rejected proof."""
    import pyref_pairing as PR
    import pyref_verifier as V
    PR.self_check(curve_id)
    c, vk, pub, proof, ch, srs_beta = restated_instance(pyref, curve_id, curve_id == 1, 7100 + curve_id)
    ok = V.open_key_for_testing(c, srs_beta)
    assert PR.PAIRINGS[curve_id].on_twist(ok["beta_h"]) and ok["beta_h"] != ok["h"]
    info = V.prepare_pcs_info(c, vk, pub, proof, ch)
    assert V.batch_verify_opening_proof(c, ok["g"], srs_beta, info) and V.batch_verify_opening_proof_pairing(c, ok, info)
    bad = V.prepare_pcs_info(c, vk, pub, dict(proof, perm_next_eval=(proof["perm_next_eval"] + 1) % c.r), ch)
    assert not V.batch_verify_opening_proof(c, ok["g"], srs_beta, bad) and not V.batch_verify_opening_proof_pairing(c, ok, bad)
    # an OpenKey of another trapdoor rejects what the right one accepts
    assert not V.batch_verify_opening_proof_pairing(c, V.open_key_for_testing(c, srs_beta + 1), info)
