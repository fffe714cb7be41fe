"""oracle/pyref_verifier.py -- TEST INFRASTRUCTURE ONLY.

Restatement of the reference VERIFIER for TurboPlonk / UltraPlonk proofs (one instance, or one aggregated BatchProof over
several), from the compressed proof bytes:
    Proof::deserialize_compressed          plonk/src/proof_system/structs.rs:59-84, 208-222, 440-450, 496-541
    Verifier::compute_challenges           plonk/src/proof_system/verifier.rs:256-321
    Verifier::compute_lin_poly_constant_term                                :340-414
    Verifier::linearization_scalars_and_bases / aggregate_poly_commitments  :421-668
    Verifier::aggregate_evaluations                                         :673-733
    Verifier::batch_verify_opening_proofs                                   :195-251
It shares no code with the restated provers (pyref_plonk / cref_prover) nor with the device path, so a proof it accepts is a
proof the reference's verification equation accepts.

The last step is a pairing check  e(A, [beta]_2) == e(B, [1]_2).  Two evaluations of it are provided: the reference's own --
the product of two pairings over the OpenKey {g, h, beta_h} (`open_key=`; oracle/pyref_pairing.py, seconds per call) -- and,
because the test SRS is generated from a known beta (as the reference's `gen_srs_for_testing` does), the equivalent
beta * A == B  in G1 (pyref.g1_* only), which most tests use for speed.

The Fiat-Shamir transcript is passed in by the caller (an object with append_message / append_vk_and_pub_input /
append_commitment(s) / append_field_elem / append_plookup_evaluations / get_and_append_challenge -- the product's
StandardTranscript, pinned by the Merlin KAT in tests/test_transcript.py).
PARITY UNPINNED by reference vectors (none exist for this path).
"""
from __future__ import annotations

import struct

import pyref as P

GATE_WIDTH = 4
PLOOKUP_EVAL_FIELDS = ("range_table_eval", "key_table_eval", "table_dom_sep_eval", "q_dom_sep_eval", "h_1_eval", "q_lookup_eval",
                       "prod_next_eval", "range_table_next_eval", "key_table_next_eval", "table_dom_sep_next_eval",
                       "h_1_next_eval", "h_2_next_eval", "q_lookup_next_eval", "w_3_next_eval", "w_4_next_eval")   # structs.rs:496-541


class VerifyError(Exception):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# ark-serialize, compressed
# ---------------------------------------------------------------------------------------------------------------------
def _sqrt_fq(c, a):
    """both base fields have q = 3 (mod 4)"""
    assert c.q % 4 == 3
    y = pow(a, (c.q + 1) // 4, c.q)
    if y * y % c.q != a % c.q:
        raise VerifyError("x is not on the curve")
    return y


def g1_decompress(c, b: bytes):
    """inverse of the G1 encodings (BLS12-381: zcash flags, big-endian; BN254: arkworks SW flags, little-endian)."""
    if c.curve_id == 0:
        if len(b) != 48 or not b[0] & 0x80:
            raise VerifyError("bad BLS12-381 G1 encoding")
        if b[0] & 0x40:
            if b[0] != 0xC0 or any(b[1:]):
                raise VerifyError("bad infinity encoding")
            return None
        greatest = bool(b[0] & 0x20)
        x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    else:
        if len(b) != 32:
            raise VerifyError("bad BN254 G1 encoding")
        if b[31] & 0x40:
            if b[31] != 0x40 or any(b[:31]):
                raise VerifyError("bad infinity encoding")
            return None
        greatest = bool(b[31] & 0x80)
        x = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little")
    if x >= c.q:
        raise VerifyError("x not reduced")
    y = _sqrt_fq(c, (x * x % c.q * x + c.b) % c.q)
    if (y > c.q - y) != greatest:
        y = c.q - y
    pt = (x, y)
    assert P.g1_on_curve(c, pt)
    return pt


class _Reader:
    def __init__(self, c, data: bytes):
        self.c, self.d, self.o = c, data, 0
        self.g1_len = 48 if c.curve_id == 0 else 32

    def take(self, n):
        if self.o + n > len(self.d):
            raise VerifyError("proof truncated")
        out = self.d[self.o:self.o + n]
        self.o += n
        return out

    def g1(self):
        return g1_decompress(self.c, self.take(self.g1_len))

    def fr(self):
        x = int.from_bytes(self.take(32), "little")
        if x >= self.c.r:
            raise VerifyError("scalar not reduced")
        return x

    def vec(self, item):
        (n,) = struct.unpack("<Q", self.take(8))
        if n > 64:
            raise VerifyError("implausible vector length")
        return [item() for _ in range(n)]


def _read_plookup(rd):
    tag = rd.take(1)[0]
    if tag == 0:
        return None
    if tag != 1:
        raise VerifyError("bad Option tag")
    pl = {"h_poly_comms": rd.vec(rd.g1), "prod_lookup_poly_comm": rd.g1()}
    pl["evals"] = {name: rd.fr() for name in PLOOKUP_EVAL_FIELDS}
    return pl


def deserialize_proof(c, data: bytes) -> dict:
    """field order of `Proof` (structs.rs:59-84)."""
    rd = _Reader(c, data)
    pr = {"wires_poly_comms": rd.vec(rd.g1), "prod_perm_poly_comm": rd.g1(), "split_quot_poly_comms": rd.vec(rd.g1),
          "opening_proof": rd.g1(), "shifted_opening_proof": rd.g1(),
          "wires_evals": rd.vec(rd.fr), "wire_sigma_evals": rd.vec(rd.fr), "perm_next_eval": rd.fr()}
    pr["plookup"] = _read_plookup(rd)
    if rd.o != len(data):
        raise VerifyError("trailing bytes")
    return pr


def deserialize_batch_proof(c, data: bytes) -> dict:
    """field order of `BatchProof` (structs.rs:266-291)."""
    rd = _Reader(c, data)
    evals = lambda: {"wires_evals": rd.vec(rd.fr), "wire_sigma_evals": rd.vec(rd.fr), "perm_next_eval": rd.fr()}
    bp = {"wires_poly_comms_vec": rd.vec(lambda: rd.vec(rd.g1)), "prod_perm_poly_comms_vec": rd.vec(rd.g1), "poly_evals_vec": rd.vec(evals),
          "plookup_proofs_vec": rd.vec(lambda: _read_plookup(rd)), "split_quot_poly_comms": rd.vec(rd.g1), "opening_proof": rd.g1(),
          "shifted_opening_proof": rd.g1()}
    if rd.o != len(data):
        raise VerifyError("trailing bytes")
    return bp


def batch_proof_from(pr: dict) -> dict:
    """impl From<Proof> for BatchProof (structs.rs:316-328)"""
    return {"wires_poly_comms_vec": [pr["wires_poly_comms"]], "prod_perm_poly_comms_vec": [pr["prod_perm_poly_comm"]],
            "poly_evals_vec": [{k: pr[k] for k in ("wires_evals", "wire_sigma_evals", "perm_next_eval")}], "plookup_proofs_vec": [pr["plookup"]],
            "split_quot_poly_comms": pr["split_quot_poly_comms"], "opening_proof": pr["opening_proof"],
            "shifted_opening_proof": pr["shifted_opening_proof"]}


# ---------------------------------------------------------------------------------------------------------------------
# the verifier
# ---------------------------------------------------------------------------------------------------------------------
def _check_lengths(vks, pubs, bp):
    k = len(bp["prod_perm_poly_comms_vec"])
    if not (len(vks) == len(pubs) == k == len(bp["wires_poly_comms_vec"]) == len(bp["poly_evals_vec"]) == len(bp["plookup_proofs_vec"])) or k == 0:
        raise VerifyError("the number of verification keys / instances / public inputs differ")


def compute_challenges_batch(transcript, vks, pubs, bp: dict, extra_msg=None) -> dict:
    """verifier.rs:256-321.  `transcript` is a fresh b"PlonkProof" transcript."""
    _check_lengths(vks, pubs, bp)
    t = transcript
    if extra_msg is not None:
        t.append_message(b"extra info", extra_msg)
    for vk, pub in zip(vks, pubs):
        t.append_vk_and_pub_input(vk["domain_size"], vk["num_inputs"], vk["k"], vk["selector_comms"], vk["sigma_comms"], pub)
    for comms in bp["wires_poly_comms_vec"]:
        t.append_commitments(b"witness_poly_comms", comms)
    ch = {"tau": t.get_and_append_challenge(b"tau")}
    for pl in bp["plookup_proofs_vec"]:
        if pl is not None:
            t.append_commitments(b"h_poly_comms", pl["h_poly_comms"])
    ch["beta"] = t.get_and_append_challenge(b"beta")
    ch["gamma"] = t.get_and_append_challenge(b"gamma")
    for cm in bp["prod_perm_poly_comms_vec"]:
        t.append_commitment(b"perm_poly_comms", cm)
    for pl in bp["plookup_proofs_vec"]:
        if pl is not None:
            t.append_commitment(b"plookup_poly_comms", pl["prod_lookup_poly_comm"])
    ch["alpha"] = t.get_and_append_challenge(b"alpha")
    t.append_commitments(b"quot_poly_comms", bp["split_quot_poly_comms"])
    ch["zeta"] = t.get_and_append_challenge(b"zeta")
    for ev in bp["poly_evals_vec"]:                                        # transcript/mod.rs:140-163
        for e in ev["wires_evals"]:
            t.append_field_elem(b"wire_evals", e)
        for e in ev["wire_sigma_evals"]:
            t.append_field_elem(b"wire_sigma_evals", e)
        t.append_field_elem(b"perm_next_eval", ev["perm_next_eval"])
    for pl in bp["plookup_proofs_vec"]:
        if pl is not None:
            t.append_plookup_evaluations(pl["evals"])
    ch["v"] = t.get_and_append_challenge(b"v")
    t.append_commitment(b"open_proof", bp["opening_proof"])
    t.append_commitment(b"shifted_open_proof", bp["shifted_opening_proof"])
    ch["u"] = t.get_and_append_challenge(b"u")
    return ch


def compute_challenges(transcript, vk: dict, pub_input, pr: dict, extra_msg=None) -> dict:
    return compute_challenges_batch(transcript, [vk], [pub_input], batch_proof_from(pr), extra_msg)


def _evaluate_pi_poly(c, n, w, pub_input, z, vanish_eval):
    """verifier.rs:845-880, unmerged circuit."""
    r = c.r
    if vanish_eval == 0:
        return 0
    vn = vanish_eval * pow(n, -1, r) % r
    out, g = 0, 1
    for val in pub_input:
        out = (out + vn * g % r * pow((z - g) % r, -1, r) % r * val) % r
        g = g * w % r
    return out


def prepare_pcs_info_batch(c, vks, pubs, bp: dict, ch: dict) -> dict:
    """verifier.rs:68-184.  Returns u, the evaluation points, the aggregated evaluation and the (scalar, base) list of the
    aggregated commitment -- over all the instances of one BatchProof."""
    r = c.r
    _check_lengths(vks, pubs, bp)
    n = vks[0]["domain_size"]
    log_n = n.bit_length() - 1
    for i, (vk, pub) in enumerate(zip(vks, pubs)):
        if len(pub) != vk["num_inputs"]:
            raise VerifyError("the circuit public input length != the %d-th verification key public input length" % i)
        if (vk.get("plookup") is not None) != (bp["plookup_proofs_vec"][i] is not None):
            raise VerifyError("Mismatched proof type and verification key type for the %d-th instance" % i)
        if vk["domain_size"] != n:
            raise VerifyError("the domain size of the %d-th verification key is different" % i)
    w = c.root_of_unity(log_n)
    w_inv = pow(w, -1, r)
    tau, beta, gamma, alpha, zeta, v, u = (ch[x] for x in ("tau", "beta", "gamma", "alpha", "zeta", "v", "u"))
    a2 = alpha * alpha % r
    a3, a4 = a2 * alpha % r, a2 * a2 % r
    a5, a6, a7 = a4 * alpha % r, a4 * a2 % r, a4 * a3 % r
    alpha_powers = [a2, a3, a4, a5, a6]
    alpha_bases = [1]                                                       # :130-138: the step is chosen by the FIRST key's type
    tmp = a7 if vks[0].get("plookup") is not None else a3
    for _ in range(len(vks) - 1):
        alpha_bases.append(tmp)
        tmp = tmp * alpha_bases[1] % r
    vanish = (pow(zeta, n, r) - 1) % r
    l1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r                       # :776-788
    ln = vanish * w_inv % r * pow(n * (zeta - w_inv) % r, -1, r) % r
    b1 = (1 + beta) % r
    g_b1 = gamma * b1 % r

    lin_const = 0
    sb = []                                                                 # (scalar, base)
    for i, (vk, pub, base) in enumerate(zip(vks, pubs, alpha_bases)):
        ev_i = bp["poly_evals_vec"][i]
        we, se, zn = ev_i["wires_evals"], ev_i["wire_sigma_evals"], ev_i["perm_next_eval"]
        pl = bp["plookup_proofs_vec"][i]
        ultra = pl is not None
        ev = pl["evals"] if ultra else None
        W = GATE_WIDTH + 1 + (1 if ultra else 0)
        if len(bp["wires_poly_comms_vec"][i]) != W or len(we) != W or len(se) != W - 1 or len(vk["sigma_comms"]) != W or len(vk["k"]) != W \
                or len(vk["selector_comms"]) != 2 * GATE_WIDTH + 5 + (1 if ultra else 0):
            raise VerifyError("wrong number of commitments / evaluations")
        # --- constant term of the linearisation polynomial (:340-414)
        tmp = (_evaluate_pi_poly(c, n, w, pub, zeta, vanish) - alpha_powers[0] * l1) % r
        acc = alpha * zn % r * ((gamma + we[W - 1]) % r) % r
        for we_i, se_i in zip(we[:W - 1], se):
            acc = acc * ((gamma + we_i + beta * se_i) % r) % r
        tmp = (tmp - acc) % r
        if ultra:
            pc = (ln * ((ev["h_1_eval"] - ev["h_2_next_eval"] - alpha_powers[0]) % r) - alpha * l1
                  - alpha_powers[1] * ((zeta - w_inv) % r) % r * ev["prod_next_eval"] % r
                  * ((g_b1 + ev["h_1_eval"] + beta * ev["h_1_next_eval"]) % r) % r * ((g_b1 + beta * ev["h_2_next_eval"]) % r)) % r
            tmp = (tmp + alpha_powers[1] * pc) % r
        lin_const = (lin_const + base * tmp) % r
        # --- [D]_1 (:513-652)
        coeff = alpha
        for we_i, k_i in zip(we, vk["k"]):
            coeff = coeff * ((beta * k_i % r * zeta + gamma + we_i) % r) % r
        coeff = (coeff + alpha_powers[0] * l1) % r * base % r
        sb.append((coeff, bp["prod_perm_poly_comms_vec"][i]))
        coeff = alpha * beta % r * zn % r
        for we_i, se_i in zip(we[:W - 1], se):
            coeff = coeff * ((beta * se_i + gamma + we_i) % r) % r
        sb.append((-coeff * base % r, vk["sigma_comms"][-1]))
        q = [we[0], we[1], we[2], we[3], we[0] * we[1] % r, we[2] * we[3] % r, pow(we[0], 5, r), pow(we[1], 5, r), pow(we[2], 5, r),
             pow(we[3], 5, r), -we[4] % r, 1, we[0] * we[1] % r * we[2] % r * we[3] % r * we[4] % r]
        for s_, comm in zip(q, vk["selector_comms"]):                       # q_lookup (14th) gets no scalar: zip stops at 13
            sb.append((s_ * base % r, comm))
        if ultra:
            merged_lookup_x = (we[5] + ev["q_lookup_eval"] * tau % r * (ev["q_dom_sep_eval"] + tau * (we[0] + tau * (we[1] + tau * we[2]))) % r) % r
            def merged_table(rng_e, key_e, ql_e, w3_e, w4_e, dom_e):         # structs.rs:925-940
                return (rng_e + ql_e * tau % r * (dom_e + tau * (key_e + tau * (w3_e + tau * w4_e))) % r) % r
            table_x = merged_table(ev["range_table_eval"], ev["key_table_eval"], ev["q_lookup_eval"], we[3], we[4], ev["table_dom_sep_eval"])
            table_xw = merged_table(ev["range_table_next_eval"], ev["key_table_next_eval"], ev["q_lookup_next_eval"], ev["w_3_next_eval"],
                                    ev["w_4_next_eval"], ev["table_dom_sep_next_eval"])
            coeff = (alpha_powers[2] * l1 + alpha_powers[3] * ln
                     + alpha_powers[4] * ((zeta - w_inv) % r) % r * b1 % r * ((gamma + merged_lookup_x) % r) % r
                     * ((g_b1 + table_x + beta * table_xw) % r)) % r
            sb.append((coeff * base % r, pl["prod_lookup_poly_comm"]))
            coeff = alpha_powers[4] * ((w_inv - zeta) % r) % r * ev["prod_next_eval"] % r * ((g_b1 + ev["h_1_eval"] + beta * ev["h_1_next_eval"]) % r) % r
            sb.append((coeff * base % r, pl["h_poly_comms"][1]))
    # split quotient commitments (:654-665)
    zeta_n2 = (1 + vanish) * zeta % r * zeta % r
    coeff = -vanish % r
    for i, cm in enumerate(bp["split_quot_poly_comms"]):
        if i:
            coeff = coeff * zeta_n2 % r
        sb.append((coeff, cm))

    # --- the remaining commitments with powers of v / u v (:453-506) and the matching evaluations (:673-733)
    eval_ = -lin_const % r
    v_base, uv_base = v, u

    def at_zeta(comm, e):
        nonlocal v_base, eval_
        sb.append((v_base, comm))
        eval_ = (eval_ + e * v_base) % r
        v_base = v_base * v % r

    def at_zeta_omega(comm, e):
        nonlocal uv_base, eval_
        sb.append((uv_base, comm))
        eval_ = (eval_ + e * uv_base) % r
        uv_base = uv_base * v % r

    # aggregate_evaluations walks the buffer in commitment order, so the two lists are zipped here
    for i, vk in enumerate(vks):
        ev_i = bp["poly_evals_vec"][i]
        wires = bp["wires_poly_comms_vec"][i]
        W = len(wires)
        pl = bp["plookup_proofs_vec"][i]
        for cm, e in zip(wires, ev_i["wires_evals"]):
            at_zeta(cm, e)
        for cm, e in zip(vk["sigma_comms"][:W - 1], ev_i["wire_sigma_evals"]):
            at_zeta(cm, e)
        at_zeta_omega(bp["prod_perm_poly_comms_vec"][i], ev_i["perm_next_eval"])
        if pl is not None:
            ev = pl["evals"]
            pvk = vk["plookup"]
            q_lookup_comm = vk["selector_comms"][-1]
            for cm, name in ((pvk["range_table_comm"], "range_table_eval"), (pvk["key_table_comm"], "key_table_eval"),
                             (pl["h_poly_comms"][0], "h_1_eval"), (q_lookup_comm, "q_lookup_eval"),
                             (pvk["table_dom_sep_comm"], "table_dom_sep_eval"), (pvk["q_dom_sep_comm"], "q_dom_sep_eval")):    # :793-805, structs.rs:545-554
                at_zeta(cm, ev[name])
            for cm, name in ((pl["prod_lookup_poly_comm"], "prod_next_eval"), (pvk["range_table_comm"], "range_table_next_eval"),
                             (pvk["key_table_comm"], "key_table_next_eval"), (pl["h_poly_comms"][0], "h_1_next_eval"),
                             (pl["h_poly_comms"][1], "h_2_next_eval"), (q_lookup_comm, "q_lookup_next_eval"),
                             (wires[3], "w_3_next_eval"), (wires[4], "w_4_next_eval"),
                             (pvk["table_dom_sep_comm"], "table_dom_sep_next_eval")):                                            # :810-826, structs.rs:557-569
                at_zeta_omega(cm, ev[name])
    return {"u": u, "eval_point": zeta, "next_eval_point": zeta * w % r, "eval": eval_, "comm_scalars_and_bases": sb,
            "opening_proof": bp["opening_proof"], "shifted_opening_proof": bp["shifted_opening_proof"]}


def prepare_pcs_info(c, vk: dict, pub_input, pr: dict, ch: dict) -> dict:
    if len(pr["split_quot_poly_comms"]) != len(pr["wires_poly_comms"]):
        raise VerifyError("wrong number of commitments / evaluations")
    return prepare_pcs_info_batch(c, [vk], [pub_input], batch_proof_from(pr), ch)


def _msm(c, pairs):
    acc = None
    for s, base in pairs:
        if base is None or s % c.r == 0:
            continue
        acc = P.g1_add(c, acc, P.g1_mul(c, s % c.r, base))
    return acc


def batch_verify_opening_proof(c, g, srs_beta: int, info: dict) -> bool:
    """verifier.rs:195-251 with one PcsInfo (r = 1).  e(A, [beta]_2) == e(B, [1]_2)  <=>  beta A == B."""
    r = c.r
    A = _msm(c, [(1, info["opening_proof"]), (info["u"], info["shifted_opening_proof"])])
    B = _msm(c, info["comm_scalars_and_bases"] + [(info["eval_point"], info["opening_proof"]),
                                                  (info["u"] * info["next_eval_point"] % r, info["shifted_opening_proof"]),
                                                  (-info["eval"] % r, g)])
    lhs = P.g1_mul(c, srs_beta % r, A) if A is not None else None
    return lhs == B


def open_key_for_testing(c, srs_beta: int) -> dict:
    """OpenKey {g, h, beta_h} of `gen_srs_for_testing` (srs.rs:118-153) with g, h the standard generators: beta_h = [beta]h."""
    import pyref_pairing as PR
    pc = PR.PAIRINGS[c.curve_id]
    return {"g": P.g1_gen(c), "h": pc.g2, "beta_h": pc.g2_mul(pc.g2, srs_beta)}


def batch_verify_opening_proof_pairing(c, open_key: dict, info: dict) -> bool:
    """verifier.rs:195-251 with one PcsInfo, as the reference evaluates it: multi_pairing([A, -B], [beta_h, h]) == 1 -- no
    trapdoor involved (oracle/pyref_pairing.py)."""
    import pyref_pairing as PR
    r = c.r
    A = _msm(c, [(1, info["opening_proof"]), (info["u"], info["shifted_opening_proof"])])
    B = _msm(c, info["comm_scalars_and_bases"] + [(info["eval_point"], info["opening_proof"]),
                                                  (info["u"] * info["next_eval_point"] % r, info["shifted_opening_proof"]),
                                                  (-info["eval"] % r, open_key["g"])])
    return PR.PAIRINGS[c.curve_id].multi_pairing_is_one([(A, open_key["beta_h"]), (P.g1_neg(c, B), open_key["h"])])


def _final_check(c, g, srs_beta, open_key, info) -> bool:
    if open_key is not None:
        return batch_verify_opening_proof_pairing(c, open_key, info)
    return batch_verify_opening_proof(c, g, srs_beta, info)


def verify(c, transcript, vk: dict, pub_input, proof_bytes: bytes, g, srs_beta, extra_msg=None, open_key=None) -> bool:
    """PlonkKzgSnark::verify (snark.rs:653-671 -> batch_verify :118-146) for one proof.
    vk: {"domain_size", "num_inputs", "k", "selector_comms", "sigma_comms", "plookup": None | {"range_table_comm",
    "key_table_comm", "table_dom_sep_comm", "q_dom_sep_comm"}} with commitments as canonical affine (x, y) or None.
    Final check: with `open_key` ({g, h, beta_h}) the reference's pairing product; otherwise its trapdoor form with
    g = powers_of_g[0] and srs_beta the trapdoor of the test SRS."""
    pr = deserialize_proof(c, proof_bytes)
    ch = compute_challenges(transcript, vk, pub_input, pr, extra_msg)
    info = prepare_pcs_info(c, vk, pub_input, pr, ch)
    return _final_check(c, g, srs_beta, open_key, info)


def verify_batch_proof(c, transcript, vks, pubs, batch_proof_bytes: bytes, g, srs_beta, extra_msg=None, open_key=None) -> bool:
    """PlonkKzgSnark::verify_batch_proof (snark.rs:148-169): one aggregated BatchProof over several instances."""
    bp = deserialize_batch_proof(c, batch_proof_bytes)
    ch = compute_challenges_batch(transcript, vks, pubs, bp, extra_msg)
    info = prepare_pcs_info_batch(c, vks, pubs, bp, ch)
    return _final_check(c, g, srs_beta, open_key, info)
