"""oracle/pyref_plonk.py -- TEST INFRASTRUCTURE ONLY.

Big-integer restatement of the arithmetic of one TurboPlonk or UltraPlonk (Plookup) proof, single instance, for
tiny circuits, following the reference's round structure:

    batch_prove_internal          plonk/src/proof_system/snark.rs:201-469
    run_1st..3rd_round            plonk/src/proof_system/prover.rs:72-209
    mask_polynomial               prover.rs:463-486      split_quotient_polynomial   prover.rs:902-960
    compute_evaluations           prover.rs:216-235      lin-poly pieces             prover.rs:302-358, 963-1035
    compute_opening_proofs        prover.rs:362-419, 490-509

Deliberately NOT the way the device computes it: the quotient is obtained by schoolbook polynomial
multiplication and exact division by X^n - 1 (no FFT, no coset), so agreement with the HIP path pins
both.  Challenges and blinding scalars are inputs (the reference's tests fix them the same way,
multiprover/proof_system/prover.rs:1316-1556); the transcript is out of scope.
PARITY UNPINNED by reference vectors (none exist); see oracle/pyref.py.
"""
from __future__ import annotations

import pyref as P


def padd(c, a, b):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % c.r for i in range(n)]


def pscale(c, a, s):
    return [x * s % c.r for x in a]


def pmul(c, a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % c.r
    return out


def pstrip(a):
    a = list(a)
    while len(a) > 1 and a[-1] == 0:
        a.pop()
    return a


def mask(c, poly, blinders, n):
    """poly + (b_0 + b_1 X + ...) * (X^n - 1)   (prover.rs:463-486)."""
    out = list(poly) + [0] * (n + len(blinders) - len(poly))
    for i, b in enumerate(blinders):
        out[i] = (out[i] - b) % c.r
        out[n + i] = (out[n + i] + b) % c.r
    return out


def div_by_vanishing(c, a, n):
    """exact quotient of a(X) / (X^n - 1); returns (quotient, remainder)."""
    a = list(a)
    q = [0] * max(len(a) - n, 0)
    for i in range(len(a) - 1, n - 1, -1):
        q[i - n] = a[i]
        a[i - n] = (a[i - n] + a[i]) % c.r
        a[i] = 0
    return q, a[:n]


def div_by_linear(c, a, z):
    """quotient of a(X) / (X - z) (remainder dropped)."""
    q = [0] * (len(a) - 1)
    carry = 0
    for k in range(len(a) - 1, 0, -1):
        carry = (a[k] + carry * z) % c.r
        q[k - 1] = carry
    return q


def sorted_lookup_vec(table, lookups):
    """merge of the lookup values into the table, in table order (constraint_system.rs:1384-1408): every table
    entry once, followed by one copy per lookup of it -- counted at the FIRST table entry holding that value."""
    counts = {}
    for e in lookups:
        counts[e] = counts.get(e, 0) + 1
    out = []
    for e in table:
        if e in counts:
            out += [e] * (1 + counts.pop(e))
        else:
            out.append(e)
    return out


def merged_table_values(c, tau, plookup, q_lookup_vals, wire_vals):
    """constraint_system.rs:1441-1461 over the whole domain."""
    r = c.r
    return [(plookup["range"][i] + q_lookup_vals[i] * tau % r * (plookup["table_dom_sep"][i] + tau * (plookup["key"][i] + tau * (
        wire_vals[3][i] + tau * wire_vals[4][i]))) % r) % r for i in range(len(q_lookup_vals))]


def merged_lookup_values(c, tau, plookup, q_lookup_vals, wire_vals):
    """constraint_system.rs:1463-1480 over the whole domain."""
    r = c.r
    return [(wire_vals[5][i] + q_lookup_vals[i] * tau % r * (plookup["q_dom_sep"][i] + tau * (wire_vals[0][i] + tau * (
        wire_vals[1][i] + tau * wire_vals[2][i]))) % r) % r for i in range(len(q_lookup_vals))]


def lookup_product_values(c, n, tau, beta, gamma, table, lookups, sorted_vec):
    """constraint_system.rs:1311-1368 before the iFFT."""
    r = c.r
    b1 = (1 + beta) % r
    g1 = gamma * b1 % r
    prod = [1]
    for j in range(n - 2):
        a = b1 * (gamma + lookups[j]) % r * (g1 + table[j] + beta * table[j + 1]) % r
        b = (g1 + sorted_vec[j] + beta * sorted_vec[j + 1]) % r * (g1 + sorted_vec[n - 1 + j] + beta * sorted_vec[n + j]) % r
        prod.append(prod[-1] * a % r * pow(b, -1, r) % r)
    prod.append(1)
    return prod


def prove_core(c, log_n, selector_vals, sigma_vals, k, wire_vals, pi_vals, blind, ch, srs_beta=None, plookup=None):
    """selector_vals: 13 x n (14 with q_lookup last for UltraPlonk), sigma_vals / wire_vals: W x n evaluations on H
    (W = 5, or 6 with Plookup); pi_vals: n.
    blind: {"wires": W x [b0,b1], "z": [b0,b1,b2], "quot": [W-1 scalars]} (+ "h": 2 x [3], "prod_lookup": [3]).
    ch: dict beta gamma alpha zeta v (+ tau).  plookup: None or {"range","key","table_dom_sep","q_dom_sep"} (n values each).
    Returns every polynomial, the evaluations and (if srs_beta) the discrete logs of the commitments."""
    r = c.r
    n = 1 << log_n
    W = len(wire_vals)
    ultra = plookup is not None
    assert W == (6 if ultra else 5) and len(selector_vals) == (14 if ultra else 13)
    w_n = c.root_of_unity(log_n)
    w_inv = pow(w_n, -1, r)
    intt = lambda vals: P.ntt_fast(c, list(vals), log_n, 1, inverse=True)
    shift = lambda poly: [cf * pow(w_n, i, r) % r for i, cf in enumerate(poly)]          # p(w X)
    sel = [intt(v) for v in selector_vals]
    sig = [intt(v) for v in sigma_vals]
    beta, gamma, alpha, zeta, v = (ch[x] for x in ("beta", "gamma", "alpha", "zeta", "v"))
    # round 1
    wire_polys = [mask(c, intt(wire_vals[i]), blind["wires"][i], n) for i in range(W)]
    pi_poly = intt(pi_vals)
    # round 1.5 (prover.rs:89-118)
    if ultra:
        tau = ch["tau"]
        tab = {x: intt(plookup[x]) for x in ("range", "key", "table_dom_sep", "q_dom_sep")}
        table = merged_table_values(c, tau, plookup, selector_vals[13], wire_vals)
        lookups = merged_lookup_values(c, tau, plookup, selector_vals[13], wire_vals)
        sorted_vec = sorted_lookup_vec(table, lookups[:n - 1])
        assert len(sorted_vec) == 2 * n - 1, "some lookup variables are outside the table"
        h_polys = [mask(c, intt(sorted_vec[:n]), blind["h"][0], n), mask(c, intt(sorted_vec[n - 1:]), blind["h"][1], n)]
    # round 2 (constraint_system.rs:1197-1223)
    prod = [1]
    for j in range(n - 1):
        a = b = 1
        for i in range(W):
            t = (wire_vals[i][j] + gamma) % r
            a = a * (t + beta * k[i] * pow(w_n, j, r)) % r
            b = b * (t + beta * sigma_vals[i][j]) % r
        prod.append(prod[-1] * a % r * pow(b, -1, r) % r)
    z_unmasked = intt(prod)
    z_poly = mask(c, z_unmasked, blind["z"], n)
    # round 2.5 (prover.rs:143-183)
    if ultra:
        pl_vals = lookup_product_values(c, n, tau, beta, gamma, table, lookups, sorted_vec)
        pl_poly = mask(c, intt(pl_vals), blind["prod_lookup"], n)
    # round 3: t = [gate + alpha*(z prod(w + beta k X + gamma) - z(wX) prod(w + beta sigma + gamma)) + alpha^2 (z - 1) L1 (+ Plookup)] / Z_H
    gate = padd(c, sel[11], pi_poly)
    for j in range(4):
        gate = padd(c, gate, pmul(c, sel[j], wire_polys[j]))
    gate = padd(c, gate, pmul(c, sel[4], pmul(c, wire_polys[0], wire_polys[1])))
    gate = padd(c, gate, pmul(c, sel[5], pmul(c, wire_polys[2], wire_polys[3])))
    ecc = wire_polys[0]
    for j in range(1, 5):
        ecc = pmul(c, ecc, wire_polys[j])
    gate = padd(c, gate, pmul(c, sel[12], ecc))
    for j in range(4):
        w2 = pmul(c, wire_polys[j], wire_polys[j])
        gate = padd(c, gate, pmul(c, sel[6 + j], pmul(c, pmul(c, w2, w2), wire_polys[j])))
    gate = padd(c, gate, pscale(c, pmul(c, sel[10], wire_polys[4]), r - 1))
    acc1 = z_poly
    acc2 = shift(z_poly)                                                       # z(w X)
    for j in range(W):
        acc1 = pmul(c, acc1, padd(c, wire_polys[j], [gamma, beta * k[j] % r]))
        acc2 = pmul(c, acc2, padd(c, padd(c, wire_polys[j], [gamma]), pscale(c, sig[j], beta)))
    perm = pscale(c, padd(c, acc1, pscale(c, acc2, r - 1)), alpha)
    l1 = intt([1] + [0] * (n - 1))                                            # L_1: 1 at w^0
    bound = pscale(c, pmul(c, padd(c, z_poly, [r - 1]), l1), alpha * alpha % r)
    numer = padd(c, padd(c, gate, perm), bound)
    if ultra:                                                                  # prover.rs:773-888
        ln = intt([0] * (n - 1) + [1])                                         # L_n: 1 at w^(n-1)
        q_lk = sel[13]

        def merged(first, dom_sep, a0, a1, a2):                                # first + q_lookup tau (dom_sep + tau (a0 + tau (a1 + tau a2)))
            inner = padd(c, a1, pscale(c, a2, tau))
            inner = padd(c, a0, pscale(c, inner, tau))
            inner = padd(c, dom_sep, pscale(c, inner, tau))
            return padd(c, first, pscale(c, pmul(c, q_lk, inner), tau))

        m_table = merged(tab["range"], tab["table_dom_sep"], tab["key"], wire_polys[3], wire_polys[4])
        m_lookup = merged(wire_polys[5], tab["q_dom_sep"], wire_polys[0], wire_polys[1], wire_polys[2])
        b1 = (1 + beta) % r
        g1 = gamma * b1 % r
        a3 = pow(alpha, 3, r)
        term_h = pmul(c, ln, padd(c, h_polys[0], pscale(c, shift(h_polys[1]), r - 1)))
        p_minus_1 = padd(c, pl_poly, [r - 1])
        left = pmul(c, pscale(c, pl_poly, b1), pmul(c, padd(c, m_lookup, [gamma]), padd(c, padd(c, m_table, [g1]), pscale(c, shift(m_table), beta))))
        right = pmul(c, shift(pl_poly), pmul(c, padd(c, padd(c, h_polys[0], [g1]), pscale(c, shift(h_polys[0]), beta)),
                                             padd(c, padd(c, h_polys[1], [g1]), pscale(c, shift(h_polys[1]), beta))))
        term_p3 = pmul(c, [(-w_inv) % r, 1], padd(c, left, pscale(c, right, r - 1)))
        numer = padd(c, numer, pscale(c, term_h, a3))
        numer = padd(c, numer, pscale(c, pmul(c, l1, p_minus_1), a3 * alpha % r))
        numer = padd(c, numer, pscale(c, pmul(c, ln, p_minus_1), a3 * alpha % r * alpha % r))
        numer = padd(c, numer, pscale(c, term_p3, pow(alpha, 6, r)))
    quot, rem = div_by_vanishing(c, numer, n)
    quot = pstrip(quot)
    divisible = not any(rem)
    # split (prover.rs:902-960)
    expected_degree = W * (n + 1) + 2
    split = [quot[i * (n + 2):(i + 1) * (n + 2)] if i < W - 1 else quot[(W - 1) * (n + 2):] for i in range(W)]
    last = 0
    for i in range(W - 1):
        now = blind["quot"][i]
        split[i] = list(split[i]) + [0] * (n + 2 - len(split[i]))
        split[i][0] = (split[i][0] - last) % r
        split[i].append(now)
        last = now
    if split[W - 1]:
        split[W - 1] = list(split[W - 1])
        split[W - 1][0] = (split[W - 1][0] - last) % r
    # round 4
    ev = lambda poly, x: P.poly_eval(c, poly, x)
    zeta_w = zeta * w_n % r
    wires_evals = [ev(p, zeta) for p in wire_polys]
    wire_sigma_evals = [ev(sig[i], zeta) for i in range(W - 1)]
    perm_next_eval = ev(z_poly, zeta_w)
    # round 5: linearisation polynomial (prover.rs:963-1035, 343-358)
    we = wires_evals
    terms = [(sel[j], we[j]) for j in range(4)]
    terms += [(sel[4], we[0] * we[1] % r), (sel[5], we[2] * we[3] % r)]
    terms += [(sel[6 + j], pow(we[j], 5, r)) for j in range(4)]
    terms += [(sel[12], we[0] * we[1] % r * we[2] % r * we[3] % r * we[4] % r), (sel[10], (-we[4]) % r), (sel[11], 1)]
    vanish = (pow(zeta, n, r) - 1) % r
    lagrange_1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r
    coeff = alpha
    for j in range(W):
        coeff = coeff * (we[j] + beta * k[j] % r * zeta + gamma) % r
    coeff = (coeff + alpha * alpha % r * lagrange_1) % r
    terms.append((z_poly, coeff))
    coeff = alpha * beta % r * perm_next_eval % r
    for j in range(W - 1):
        coeff = coeff * (we[j] + beta * wire_sigma_evals[j] + gamma) % r
    terms.append((sig[W - 1], (-coeff) % r))
    plookup_evals = None
    if ultra:                                                                  # prover.rs:238-299, 1037-1112
        pe = {"range_table_eval": ev(tab["range"], zeta), "key_table_eval": ev(tab["key"], zeta), "h_1_eval": ev(h_polys[0], zeta),
              "q_lookup_eval": ev(sel[13], zeta), "prod_next_eval": ev(pl_poly, zeta_w), "table_dom_sep_eval": ev(tab["table_dom_sep"], zeta),
              "q_dom_sep_eval": ev(tab["q_dom_sep"], zeta), "range_table_next_eval": ev(tab["range"], zeta_w),
              "key_table_next_eval": ev(tab["key"], zeta_w), "h_1_next_eval": ev(h_polys[0], zeta_w), "h_2_next_eval": ev(h_polys[1], zeta_w),
              "q_lookup_next_eval": ev(sel[13], zeta_w), "w_3_next_eval": ev(wire_polys[3], zeta_w), "w_4_next_eval": ev(wire_polys[4], zeta_w),
              "table_dom_sep_next_eval": ev(tab["table_dom_sep"], zeta_w)}
        plookup_evals = pe
        em = lambda first, ql, ds, a0, a1, a2: (first + ql * tau % r * (ds + tau * (a0 + tau * (a1 + tau * a2))) % r) % r
        mt = em(pe["range_table_eval"], pe["q_lookup_eval"], pe["table_dom_sep_eval"], pe["key_table_eval"], we[3], we[4])
        mt_next = em(pe["range_table_next_eval"], pe["q_lookup_next_eval"], pe["table_dom_sep_next_eval"], pe["key_table_next_eval"],
                     pe["w_3_next_eval"], pe["w_4_next_eval"])
        ml = em(we[5], pe["q_lookup_eval"], pe["q_dom_sep_eval"], we[0], we[1], we[2])
        lagrange_n = vanish * w_inv % r * pow(n * (zeta - w_inv) % r, -1, r) % r
        a4, a5, a6 = (pow(alpha, e, r) for e in (4, 5, 6))
        zmg = (zeta - w_inv) % r
        coeff = (a4 * lagrange_1 + a5 * lagrange_n + a6 * zmg % r * b1 % r * ((gamma + ml) % r) % r * ((g1 + mt + beta * mt_next) % r)) % r
        terms.append((pl_poly, coeff))
        coeff = a6 * zmg % r * pe["prod_next_eval"] % r * ((g1 + pe["h_1_eval"] + beta * pe["h_1_next_eval"]) % r) % r
        terms.append((h_polys[1], (-coeff) % r))
    zeta_n2 = (vanish + 1) * zeta % r * zeta % r
    cf = 1
    for i in range(W):
        terms.append((split[i], (-vanish) * cf % r))
        cf = cf * zeta_n2 % r
    lin = [0]
    for poly, s in terms:
        lin = padd(c, lin, pscale(c, poly, s))
    # opening proofs (prover.rs:362-419, 421-460, 490-509)
    open_polys = [lin] + wire_polys + sig[:W - 1]
    shifted_polys = [z_poly]
    if ultra:
        open_polys += [tab["range"], tab["key"], h_polys[0], sel[13], tab["table_dom_sep"], tab["q_dom_sep"]]
        shifted_polys += [pl_poly, tab["range"], tab["key"], h_polys[0], h_polys[1], sel[13], wire_polys[3], wire_polys[4], tab["table_dom_sep"]]

    def batched(polys, point):
        batch, cf = [0], 1
        for poly in polys:
            batch = padd(c, batch, pscale(c, poly, cf))
            cf = cf * v % r
        return div_by_linear(c, batch, point)

    opening = batched(open_polys, zeta)
    shifted = batched(shifted_polys, zeta_w)
    out = {"wire_polys": wire_polys, "pi_poly": pi_poly, "z_poly": z_poly, "quot": quot, "divisible": divisible,
           "quot_degree_ok": len(quot) - 1 == expected_degree, "split": split,
           "wires_evals": wires_evals, "wire_sigma_evals": wire_sigma_evals, "perm_next_eval": perm_next_eval,
           "lin_poly": lin, "opening_poly": opening, "shifted_opening_poly": shifted, "selectors": sel, "sigmas": sig}
    if ultra:
        out.update({"merged_table": table, "merged_lookups": lookups, "sorted_vec": sorted_vec, "h_polys": h_polys,
                    "prod_lookup_values": pl_vals, "prod_lookup_poly": pl_poly, "plookup_evals": plookup_evals, "table_polys": tab})
    if srs_beta is not None:
        dl = lambda poly: P.poly_eval(c, poly, srs_beta)
        out["commit_dlogs"] = {"wires": [dl(p) for p in wire_polys], "z": dl(z_poly), "split": [dl(p) for p in split],
                               "opening": dl(opening), "shifted_opening": dl(shifted)}
        if ultra:
            out["commit_dlogs"].update({"h": [dl(p) for p in h_polys], "prod_lookup": dl(pl_poly)})
    return out


def batch_prove_core(c, log_n, instances, ch, quot_blind, srs_beta=None):
    """Aggregated proof over several instances of one domain size -- `batch_prove_internal` (snark.rs:201-469) assembled from
    `prove_core` runs of the single instances, which is legitimate because everything shared is linear in them:
      quotient      t = sum_k alpha_base_k t_k, alpha_base_{k+1} = alpha_base_k * alpha^3 (alpha^7 with Plookup)  (prover.rs:661-669)
      lin. poly     the quotient part once (prover.rs:343-358) + alpha_base_k * (non-quotient part of instance k)  (snark.rs:403-428)
      openings      one batched witness polynomial per point over the concatenated lists                            (prover.rs:362-419)
    instances: dicts {"selector_vals", "sigma_vals", "k", "wire_vals", "pi_vals", "blind", "plookup"}; the "quot" entry of a blind is
    ignored -- quot_blind holds the W - 1 scalars of the one split.  ch: tau beta gamma alpha zeta v, shared."""
    r = c.r
    n = 1 << log_n
    w_n = c.root_of_unity(log_n)
    alpha, zeta, v = ch["alpha"], ch["zeta"], ch["v"]
    W = len(instances[0]["wire_vals"])
    outs, bases, base = [], [], 1
    for inst in instances:
        assert len(inst["wire_vals"]) == W, "inconsistent plonk circuit types"
        blind = dict(inst["blind"], quot=[0] * (W - 1))
        outs.append(prove_core(c, log_n, inst["selector_vals"], inst["sigma_vals"], inst["k"], inst["wire_vals"], inst["pi_vals"], blind, ch,
                               None, plookup=inst.get("plookup")))
        bases.append(base)
        base = base * pow(alpha, 7 if inst.get("plookup") is not None else 3, r) % r
    vanish = (pow(zeta, n, r) - 1) % r
    zeta_n2 = (vanish + 1) * zeta % r * zeta % r

    def quotient_lin_part(split):
        acc, cf = [0], 1
        for p in split:
            acc = padd(c, acc, pscale(c, p, (-vanish) * cf % r))
            cf = cf * zeta_n2 % r
        return acc

    quot = [0]
    for o, b in zip(outs, bases):
        quot = padd(c, quot, pscale(c, o["quot"], b))
    quot = pstrip(quot)
    split = [quot[i * (n + 2):(i + 1) * (n + 2)] if i < W - 1 else quot[(W - 1) * (n + 2):] for i in range(W)]
    last = 0
    for i in range(W - 1):
        now = quot_blind[i]
        split[i] = list(split[i]) + [0] * (n + 2 - len(split[i]))
        split[i][0] = (split[i][0] - last) % r
        split[i].append(now)
        last = now
    if split[W - 1]:
        split[W - 1] = list(split[W - 1])
        split[W - 1][0] = (split[W - 1][0] - last) % r
    lin = quotient_lin_part(split)
    for o, b in zip(outs, bases):
        non_quot = padd(c, o["lin_poly"], pscale(c, quotient_lin_part(o["split"]), r - 1))
        lin = padd(c, lin, pscale(c, non_quot, b))
    open_polys, shifted_polys = [lin], []
    for o in outs:
        Wk = len(o["wire_polys"])
        open_polys += o["wire_polys"] + o["sigmas"][:Wk - 1]
        shifted_polys.append(o["z_poly"])
        if "h_polys" in o:
            tab, sel, h, pl = o["table_polys"], o["selectors"], o["h_polys"], o["prod_lookup_poly"]
            open_polys += [tab["range"], tab["key"], h[0], sel[13], tab["table_dom_sep"], tab["q_dom_sep"]]
            shifted_polys += [pl, tab["range"], tab["key"], h[0], h[1], sel[13], o["wire_polys"][3], o["wire_polys"][4], tab["table_dom_sep"]]

    def batched(polys, point):
        batch, cf = [0], 1
        for poly in polys:
            batch = padd(c, batch, pscale(c, poly, cf))
            cf = cf * v % r
        return div_by_linear(c, batch, point)

    opening = batched(open_polys, zeta)
    shifted = batched(shifted_polys, zeta * w_n % r)
    res = {"instances": outs, "alpha_bases": bases, "quot": quot, "split": split, "lin_poly": lin, "opening_poly": opening,
           "shifted_opening_poly": shifted, "divisible": all(o["divisible"] for o in outs),
           "quot_degree_ok": len(quot) - 1 == W * (n + 1) + 2}
    if srs_beta is not None:
        dl = lambda poly: P.poly_eval(c, poly, srs_beta)
        res["commit_dlogs"] = {"split": [dl(p) for p in split], "opening": dl(opening), "shifted_opening": dl(shifted),
                               "wires": [[dl(p) for p in o["wire_polys"]] for o in outs], "z": [dl(o["z_poly"]) for o in outs],
                               "h": [[dl(p) for p in o["h_polys"]] if "h_polys" in o else None for o in outs],
                               "prod_lookup": [dl(o["prod_lookup_poly"]) if "h_polys" in o else None for o in outs]}
    return res
