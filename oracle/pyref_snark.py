"""oracle/pyref_snark.py -- TEST INFRASTRUCTURE ONLY.

`PlonkKzgSnark::prove` (plonk/src/proof_system/snark.rs:201-469, one instance) restated around the schoolbook prover
`pyref_plonk.prove_core`: the challenges come from the Fiat-Shamir transcript, round after round.  prove_core takes all the
challenges at once, but what it computes for round k depends on the challenges of the earlier rounds only -- so it is simply
re-run with the challenges known so far (fixed generic values for the rest) until every round's commitments have been absorbed.  Commitments
go through the trapdoor (commit(p) = [p(beta)]G); the bytes follow `Proof::serialize_compressed` (structs.rs:59-84).

The transcript object and the two encoders (g1_bytes, fr_bytes) are passed in by the caller -- the product's pure-Python
transcript module, pinned by the Merlin / Keccak KATs of tests/test_transcript.py -- so that nothing here imports the product.
Generates tests/golden/proof_vectors.json (tests/golden/make_proof_golden.py); tiny circuits only (schoolbook arithmetic).
PARITY UNPINNED by reference vectors (none exist).
"""
from __future__ import annotations

import struct

import pyref as P
import pyref_plonk as PP

PLOOKUP_EVAL_FIELDS = ("range_table_eval", "key_table_eval", "table_dom_sep_eval", "q_dom_sep_eval", "h_1_eval", "q_lookup_eval",
                       "prod_next_eval", "range_table_next_eval", "key_table_next_eval", "table_dom_sep_next_eval",
                       "h_1_next_eval", "h_2_next_eval", "q_lookup_next_eval", "w_3_next_eval", "w_4_next_eval")   # structs.rs:496-541


def prove(c, log_n, selector_vals, sigma_vals, k, wire_vals, pi_vals, pub_input, blind, srs_beta, transcript, g1_bytes, fr_bytes,
          plookup=None, extra_msg=None, srs_g=None):
    """Returns {"proof": compressed Proof bytes, "vk": verifying-key commitments (affine points), "challenges", "core": prove_core output}.
    srs_g: the SRS is powers_of_g[i] = srs_beta^i * srs_g (default: the curve's standard generator; universal_setup_for_testing
    draws a random one, snark.rs:496)."""
    r = c.r
    n = 1 << log_n
    G = P.g1_gen(c) if srs_g is None else srs_g
    ultra = plookup is not None
    pt = lambda dlog: P.g1_mul(c, dlog % r, G) if dlog % r else None
    # placeholders for the rounds not reached yet (generic values: no accidental zero denominators)
    ch = {x: 0x9e3779b97f4a7c15f39cc0605cedc835 + 0x1000003 * i for i, x in enumerate(("tau", "beta", "gamma", "alpha", "zeta", "v"))}
    run = lambda: PP.prove_core(c, log_n, selector_vals, sigma_vals, k, wire_vals, pi_vals, blind, ch, srs_beta, plookup=plookup)
    out = run()
    commit = lambda poly: pt(P.poly_eval(c, poly, srs_beta))
    vk = {"domain_size": n, "num_inputs": len(pub_input), "k": list(k), "selector_comms": [commit(p) for p in out["selectors"]],
          "sigma_comms": [commit(p) for p in out["sigmas"]], "plookup": None}
    if ultra:
        tab = out["table_polys"]
        vk["plookup"] = {"range_table_comm": commit(tab["range"]), "key_table_comm": commit(tab["key"]),
                         "table_dom_sep_comm": commit(tab["table_dom_sep"]), "q_dom_sep_comm": commit(tab["q_dom_sep"])}
    t = transcript
    if extra_msg is not None:
        t.append_message(b"extra info", extra_msg)
    t.append_vk_and_pub_input(n, len(pub_input), k, vk["selector_comms"], vk["sigma_comms"], pub_input)
    # round 1 / 1.5 (snark.rs:277-323)
    wires = [pt(d) for d in out["commit_dlogs"]["wires"]]
    t.append_commitments(b"witness_poly_comms", wires)
    ch["tau"] = t.get_and_append_challenge(b"tau")
    h = None
    if ultra:
        out = run()
        h = [pt(d) for d in out["commit_dlogs"]["h"]]
        t.append_commitments(b"h_poly_comms", h)
    # round 2 / 2.5
    ch["beta"] = t.get_and_append_challenge(b"beta")
    ch["gamma"] = t.get_and_append_challenge(b"gamma")
    out = run()
    z = pt(out["commit_dlogs"]["z"])
    t.append_commitment(b"perm_poly_comms", z)
    pl = None
    if ultra:
        pl = pt(out["commit_dlogs"]["prod_lookup"])
        t.append_commitment(b"plookup_poly_comms", pl)
    # round 3
    ch["alpha"] = t.get_and_append_challenge(b"alpha")
    out = run()
    assert out["divisible"] and out["quot_degree_ok"], "the witness does not satisfy the circuit"
    split = [pt(d) for d in out["commit_dlogs"]["split"]]
    t.append_commitments(b"quot_poly_comms", split)
    # round 4 / 4.5
    ch["zeta"] = t.get_and_append_challenge(b"zeta")
    out = run()
    for e in out["wires_evals"]:
        t.append_field_elem(b"wire_evals", e)
    for e in out["wire_sigma_evals"]:
        t.append_field_elem(b"wire_sigma_evals", e)
    t.append_field_elem(b"perm_next_eval", out["perm_next_eval"])
    if ultra:
        t.append_plookup_evaluations(out["plookup_evals"])
    # round 5
    ch["v"] = t.get_and_append_challenge(b"v")
    out = run()
    opening, shifted = pt(out["commit_dlogs"]["opening"]), pt(out["commit_dlogs"]["shifted_opening"])
    vec = lambda items, enc: struct.pack("<Q", len(items)) + b"".join(enc(x) for x in items)
    blob = vec(wires, g1_bytes) + g1_bytes(z) + vec(split, g1_bytes) + g1_bytes(opening) + g1_bytes(shifted)
    blob += vec(out["wires_evals"], fr_bytes) + vec(out["wire_sigma_evals"], fr_bytes) + fr_bytes(out["perm_next_eval"])
    if ultra:
        blob += b"\x01" + vec(h, g1_bytes) + g1_bytes(pl) + b"".join(fr_bytes(out["plookup_evals"][name]) for name in PLOOKUP_EVAL_FIELDS)
    else:
        blob += b"\x00"
    return {"proof": blob, "vk": vk, "challenges": dict(ch), "core": out}


def batch_prove(c, log_n, instances, pub_inputs, quot_blind, srs_beta, transcript, g1_bytes, fr_bytes, extra_msg=None):
    """`PlonkKzgSnark::batch_prove` (snark.rs:64-78, 201-469) restated around pyref_plonk.batch_prove_core, by the same re-run
    scheme as `prove`.  instances: the dicts batch_prove_core takes.  Returns {"proof": compressed BatchProof bytes
    (structs.rs:266-291), "vks", "challenges"}."""
    r = c.r
    n = 1 << log_n
    G = P.g1_gen(c)
    pt = lambda dlog: P.g1_mul(c, dlog % r, G) if dlog % r else None
    commit = lambda poly: pt(P.poly_eval(c, poly, srs_beta))
    ch = {x: 0x9e3779b97f4a7c15f39cc0605cedc835 + 0x1000003 * i for i, x in enumerate(("tau", "beta", "gamma", "alpha", "zeta", "v"))}
    run = lambda: PP.batch_prove_core(c, log_n, instances, ch, quot_blind, srs_beta)
    out = run()
    K = len(instances)
    ultra = [inst.get("plookup") is not None for inst in instances]
    vks = []
    for inst, o, u in zip(instances, out["instances"], ultra):
        vk = {"domain_size": n, "num_inputs": 0, "k": list(inst["k"]), "selector_comms": [commit(p) for p in o["selectors"]],
              "sigma_comms": [commit(p) for p in o["sigmas"]], "plookup": None}
        if u:
            tab = o["table_polys"]
            vk["plookup"] = {"range_table_comm": commit(tab["range"]), "key_table_comm": commit(tab["key"]),
                             "table_dom_sep_comm": commit(tab["table_dom_sep"]), "q_dom_sep_comm": commit(tab["q_dom_sep"])}
        vks.append(vk)
    t = transcript
    if extra_msg is not None:
        t.append_message(b"extra info", extra_msg)
    for vk, pub in zip(vks, pub_inputs):
        vk["num_inputs"] = len(pub)
        t.append_vk_and_pub_input(n, len(pub), vk["k"], vk["selector_comms"], vk["sigma_comms"], pub)
    dl = out["commit_dlogs"]
    wires = [[pt(d) for d in ws] for ws in dl["wires"]]
    for ws in wires:
        t.append_commitments(b"witness_poly_comms", ws)
    ch["tau"] = t.get_and_append_challenge(b"tau")
    dl = run()["commit_dlogs"]
    h = [[pt(d) for d in hs] if hs is not None else None for hs in dl["h"]]
    for hs in h:
        if hs is not None:
            t.append_commitments(b"h_poly_comms", hs)
    ch["beta"] = t.get_and_append_challenge(b"beta")
    ch["gamma"] = t.get_and_append_challenge(b"gamma")
    dl = run()["commit_dlogs"]
    z = [pt(d) for d in dl["z"]]
    for cm in z:
        t.append_commitment(b"perm_poly_comms", cm)
    pl = [pt(d) if d is not None else None for d in dl["prod_lookup"]]
    for cm, u in zip(pl, ultra):
        if u:
            t.append_commitment(b"plookup_poly_comms", cm)
    ch["alpha"] = t.get_and_append_challenge(b"alpha")
    out = run()
    assert out["divisible"] and out["quot_degree_ok"]
    split = [pt(d) for d in out["commit_dlogs"]["split"]]
    t.append_commitments(b"quot_poly_comms", split)
    ch["zeta"] = t.get_and_append_challenge(b"zeta")
    out = run()
    for o in out["instances"]:
        for e in o["wires_evals"]:
            t.append_field_elem(b"wire_evals", e)
        for e in o["wire_sigma_evals"]:
            t.append_field_elem(b"wire_sigma_evals", e)
        t.append_field_elem(b"perm_next_eval", o["perm_next_eval"])
    for o, u in zip(out["instances"], ultra):
        if u:
            t.append_plookup_evaluations(o["plookup_evals"])
    ch["v"] = t.get_and_append_challenge(b"v")
    out = run()
    dl = out["commit_dlogs"]
    vec = lambda items, enc: struct.pack("<Q", len(items)) + b"".join(enc(x) for x in items)
    blob = vec(wires, lambda ws: vec(ws, g1_bytes)) + vec(z, g1_bytes)
    blob += vec(out["instances"], lambda o: vec(o["wires_evals"], fr_bytes) + vec(o["wire_sigma_evals"], fr_bytes) + fr_bytes(o["perm_next_eval"]))

    def plookup(i):
        if not ultra[i]:
            return b"\x00"
        ev = out["instances"][i]["plookup_evals"]
        return b"\x01" + vec(h[i], g1_bytes) + g1_bytes(pl[i]) + b"".join(fr_bytes(ev[name]) for name in PLOOKUP_EVAL_FIELDS)

    blob += vec(list(range(K)), plookup)
    blob += vec(split, g1_bytes) + g1_bytes(pt(dl["opening"])) + g1_bytes(pt(dl["shifted_opening"]))
    return {"proof": blob, "vks": vks, "challenges": dict(ch)}
