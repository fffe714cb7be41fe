"""oracle/cref_prover.py -- TEST INFRASTRUCTURE ONLY.

One TurboPlonk proof on the CPU, assembled from the C restatement (oracle/cpu_ref.c: ark-poly style radix-2 FFTs, the
quotient closure of prover.rs:605-659, the serial grand product, ark-ec style Pippenger) in the order of
`PlonkKzgSnark::batch_prove_internal` (plonk/src/proof_system/snark.rs:201-469) -- the CPU side of config C1
(BASELINE.json: "TurboPlonk over BLS12-381, 2^10 constraints, CPU-only").  Unlike pyref_plonk (schoolbook, n <= 64) it runs
at 2^10..2^13 gates in seconds, so the device prover can be checked at a size where every NTT is multi-pass and every MSM
runs on the precomputed-table path.  Scalars (challenges, blinders) are inputs, as in pyref_plonk.
PARITY UNPINNED by reference vectors (none exist); its pieces are pinned in tests/test_oracle.py.
"""
from __future__ import annotations

import time

import numpy as np

import cref


def _mont(curve, r, vals):
    return cref.fr_convert(curve, cref.ints_to_limbs([v % r for v in vals], 4), True)


def _ints(curve, limbs):
    return cref.limbs_to_ints(cref.fr_convert(curve, np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 4), False))


def prove_turbo(curve: int, r: int, fr_generator: int, log_n: int, selector_vals, sigma_vals, k, wire_vals, pi_vals, blind, ch, srs_xy, threads: int = 1):
    """selector_vals (13, n, 4), sigma_vals / wire_vals (5, n, 4), pi_vals (n, 4): Montgomery evaluations on H.
    blind: {"wires": 5 x [b0, b1], "z": [3], "quot": [4]} ints; ch: {"beta", "gamma", "alpha", "zeta", "v"} ints;
    srs_xy: (>= n + 3, 2, fq_limbs) affine powers of g.  Returns commitments (affine limbs), evaluations (ints) and seconds."""
    t_start = time.perf_counter()
    n = 1 << log_n
    W = 5
    spent = {"ntt": 0.0, "msm": 0.0, "quotient_round": 0.0, "grand_product": 0.0}

    def timed(key, fn, *a, **kw):
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        spent[key] += time.perf_counter() - t0
        return out
    ntt = lambda a, inverse, coset=None, lg=log_n: timed("ntt", cref.ntt, curve, a, lg, inverse, coset, threads=threads)
    k_m = _mont(curve, r, k)
    beta, gamma, alpha, zeta, v = (ch[x] for x in ("beta", "gamma", "alpha", "zeta", "v"))
    bm, gm, am = (_mont(curve, r, [x])[0] for x in (beta, gamma, alpha))
    sel = np.stack([ntt(selector_vals[i], True) for i in range(13)])
    sig = np.stack([ntt(sigma_vals[i], True) for i in range(W)])
    t_prove = time.perf_counter()                                        # what precedes is `preprocess` (snark.rs:529-617), not `prove`

    def mask(poly, b):                                                   # prover.rs:463-486
        out = np.zeros((n + len(b), 4), dtype=np.uint64)
        out[:n] = poly
        head = _ints(curve, out[:len(b)])
        out[:len(b)] = _mont(curve, r, [(h - x) % r for h, x in zip(head, b)])
        out[n:] = _mont(curve, r, b)
        return out

    commit = lambda p: cref.jac_to_affine(curve, timed("msm", cref.msm, curve, srs_xy[:p.shape[0]], p, scalars_are_mont=True, threads=threads))[0]
    # round 1
    wire_polys = [mask(ntt(wire_vals[i], True), blind["wires"][i]) for i in range(W)]
    pi_poly = ntt(pi_vals, True)
    wires_comms = [commit(p) for p in wire_polys]
    # round 2 (constraint_system.rs:1197-1223)
    z_poly = mask(timed("grand_product", cref.plonk_perm_product, curve, log_n, np.stack(wire_vals), np.stack(sigma_vals), k_m, bm, gm, threads=threads),
                  blind["z"])
    z_comm = commit(z_poly)
    # round 3 (prover.rs:512-673, 902-960)
    slab = np.zeros((25, n + 3, 4), dtype=np.uint64)
    slab[:13, :n] = sel
    slab[13:18, :n] = sig
    for i in range(W):
        slab[18 + i, :n + 2] = wire_polys[i]
    slab[23] = z_poly
    slab[24, :n] = pi_poly
    quot = timed("quotient_round", cref.plonk_quotient, curve, log_n, slab, k_m, am, bm, gm, threads=threads)    # 25 coset FFTs + closure + coset iFFT
    expected = W * (n + 1) + 2
    assert not quot[expected + 1:].any() and quot[expected].any(), "quotient degree (prover.rs:916-919)"
    split, last = [], 0
    for i in range(W):
        lo = i * (n + 2)
        hi = (i + 1) * (n + 2) if i < W - 1 else expected + 1
        p = np.zeros((n + 3, 4), dtype=np.uint64)
        p[:hi - lo] = quot[lo:hi]
        if i < W - 1:
            p[n + 2] = _mont(curve, r, [blind["quot"][i]])[0]
        if last:
            p[0] = _mont(curve, r, [(_ints(curve, p[:1])[0] - last) % r])[0]
        last = blind["quot"][i] if i < W - 1 else 0
        split.append(p if i < W - 1 else p[:hi - lo])
    split_comms = [commit(p) for p in split]
    # round 4 (prover.rs:216-235)
    w_n = pow(fr_generator, (r - 1) >> log_n, r)
    ev = lambda p, x: _ints(curve, cref.poly_eval(curve, p, _mont(curve, r, [x])[0]))[0]
    we = [ev(p, zeta) for p in wire_polys]
    se = [ev(sig[i], zeta) for i in range(W - 1)]
    perm_next = ev(z_poly, zeta * w_n % r)
    # round 5 (prover.rs:302-358, 963-1035, 362-419, 490-509)
    terms = [(sel[j], we[j]) for j in range(4)] + [(sel[4], we[0] * we[1] % r), (sel[5], we[2] * we[3] % r)]
    terms += [(sel[6 + j], pow(we[j], 5, r)) for j in range(4)]
    terms += [(sel[12], we[0] * we[1] * we[2] * we[3] * we[4] % r), (sel[10], (-we[4]) % r), (sel[11], 1)]
    vanish = (pow(zeta, n, r) - 1) % r
    lagrange_1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r
    cf = alpha
    for j in range(W):
        cf = cf * (we[j] + beta * k[j] % r * zeta + gamma) % r
    terms.append((z_poly, (cf + alpha * alpha % r * lagrange_1) % r))
    cf = alpha * beta % r * perm_next % r
    for j in range(W - 1):
        cf = cf * (we[j] + beta * se[j] + gamma) % r
    terms.append((sig[W - 1], (-cf) % r))
    zeta_n2 = (vanish + 1) * zeta % r * zeta % r
    cf = 1
    for i in range(W):
        terms.append((split[i], (-vanish) * cf % r))
        cf = cf * zeta_n2 % r
    lincomb = lambda ts: cref.poly_lincomb(curve, [p for p, _ in ts], _mont(curve, r, [s for _, s in ts]), n + 3)
    lin = lincomb(terms)
    bt, cf = [], 1
    for p in [lin] + wire_polys + [sig[i] for i in range(W - 1)]:
        bt.append((p, cf))
        cf = cf * v % r
    opening = cref.poly_div_linear(curve, lincomb(bt), _mont(curve, r, [zeta])[0])
    shifted = cref.poly_div_linear(curve, z_poly, _mont(curve, r, [zeta * w_n % r])[0])
    return {"wires_comms": wires_comms, "z_comm": z_comm, "split_comms": split_comms, "opening": commit(opening), "shifted": commit(shifted),
            "wires_evals": we, "wire_sigma_evals": se, "perm_next_eval": perm_next, "seconds": time.perf_counter() - t_start,
            "prove_seconds": time.perf_counter() - t_prove,
            "spent_seconds": {k: round(v, 3) for k, v in spent.items()}}


def prove_ultra(curve: int, r: int, fr_generator: int, log_n: int, selector_vals, sigma_vals, table_vals, k, wire_vals, pi_vals, blind, ch, srs_xy, threads: int = 1):
    """UltraPlonk (Plookup) counterpart of prove_turbo: selector_vals (14, n, 4) with q_lookup last, sigma_vals / wire_vals (6, n, 4),
    table_vals (4, n, 4) = range, key, table_dom_sep, q_dom_sep; blind additionally {"h": 2 x [3], "prod_lookup": [3]}, ch additionally "tau".
    Rounds 1.5 / 2.5 / 4.5 and the Plookup parts of rounds 3 and 5 follow prover.rs:89-183, 238-299, 421-460, 773-888, 1037-1112."""
    t_start = time.perf_counter()
    n = 1 << log_n
    W = 6
    ntt = lambda a, inverse: cref.ntt(curve, a, log_n, inverse, None, threads=threads)
    k_m = _mont(curve, r, k)
    tau, beta, gamma, alpha, zeta, v = (ch[x] for x in ("tau", "beta", "gamma", "alpha", "zeta", "v"))
    tm, bm, gm, am = (_mont(curve, r, [x])[0] for x in (tau, beta, gamma, alpha))
    sel = np.stack([ntt(selector_vals[i], True) for i in range(14)])
    sig = np.stack([ntt(sigma_vals[i], True) for i in range(W)])
    tab = np.stack([ntt(table_vals[i], True) for i in range(4)])

    def mask(poly, b):
        out = np.zeros((n + len(b), 4), dtype=np.uint64)
        out[:n] = poly
        head = _ints(curve, out[:len(b)])
        out[:len(b)] = _mont(curve, r, [(h - x) % r for h, x in zip(head, b)])
        out[n:] = _mont(curve, r, b)
        return out

    commit = lambda p: cref.jac_to_affine(curve, cref.msm(curve, srs_xy[:p.shape[0]], p, scalars_are_mont=True, threads=threads))[0]
    wire_polys = [mask(ntt(wire_vals[i], True), blind["wires"][i]) for i in range(W)]
    pi_poly = ntt(pi_vals, True)
    wires_comms = [commit(p) for p in wire_polys]
    # round 1.5
    table, lookup = cref.plookup_merge(curve, np.stack(wire_vals), np.stack(table_vals), selector_vals[13], tm)
    sorted_vec = cref.plookup_sorted(curve, table, lookup)
    assert sorted_vec is not None and sorted_vec.shape[0] == 2 * n - 1, "some lookup variables might be outside the table"
    h_polys = [mask(ntt(sorted_vec[:n], True), blind["h"][0]), mask(ntt(sorted_vec[n - 1:], True), blind["h"][1])]
    h_comms = [commit(p) for p in h_polys]
    # round 2, 2.5
    z_poly = mask(cref.plonk_perm_product(curve, log_n, np.stack(wire_vals), np.stack(sigma_vals), k_m, bm, gm, threads=threads), blind["z"])
    z_comm = commit(z_poly)
    pl_poly = mask(cref.plookup_product(curve, log_n, table, lookup, sorted_vec, bm, gm, threads=threads), blind["prod_lookup"])
    pl_comm = commit(pl_poly)
    # round 3
    slab = np.zeros((35, n + 3, 4), dtype=np.uint64)
    slab[:14, :n] = sel
    slab[14:20, :n] = sig
    slab[20:24, :n] = tab
    for i in range(W):
        slab[24 + i, :n + 2] = wire_polys[i]
    slab[30] = z_poly
    slab[31, :n] = pi_poly
    slab[32], slab[33], slab[34] = h_polys[0], h_polys[1], pl_poly
    quot = cref.plonk_quotient_ultra(curve, log_n, slab, k_m, tm, am, bm, gm, threads=threads)
    expected = W * (n + 1) + 2
    assert not quot[expected + 1:].any() and quot[expected].any(), "quotient degree (prover.rs:916-919)"
    split, last = [], 0
    for i in range(W):
        lo = i * (n + 2)
        hi = (i + 1) * (n + 2) if i < W - 1 else expected + 1
        p = np.zeros((n + 3, 4), dtype=np.uint64)
        p[:hi - lo] = quot[lo:hi]
        if i < W - 1:
            p[n + 2] = _mont(curve, r, [blind["quot"][i]])[0]
        if last:
            p[0] = _mont(curve, r, [(_ints(curve, p[:1])[0] - last) % r])[0]
        last = blind["quot"][i] if i < W - 1 else 0
        split.append(p if i < W - 1 else p[:hi - lo])
    split_comms = [commit(p) for p in split]
    # round 4, 4.5
    w_n = pow(fr_generator, (r - 1) >> log_n, r)
    zeta_w = zeta * w_n % r
    ev = lambda p, x: _ints(curve, cref.poly_eval(curve, p, _mont(curve, r, [x])[0]))[0]
    we = [ev(p, zeta) for p in wire_polys]
    se = [ev(sig[i], zeta) for i in range(W - 1)]
    perm_next = ev(z_poly, zeta_w)
    pe = {"range_table_eval": ev(tab[0], zeta), "key_table_eval": ev(tab[1], zeta), "table_dom_sep_eval": ev(tab[2], zeta), "q_dom_sep_eval": ev(tab[3], zeta),
          "h_1_eval": ev(h_polys[0], zeta), "q_lookup_eval": ev(sel[13], zeta), "prod_next_eval": ev(pl_poly, zeta_w),
          "range_table_next_eval": ev(tab[0], zeta_w), "key_table_next_eval": ev(tab[1], zeta_w), "table_dom_sep_next_eval": ev(tab[2], zeta_w),
          "h_1_next_eval": ev(h_polys[0], zeta_w), "h_2_next_eval": ev(h_polys[1], zeta_w), "q_lookup_next_eval": ev(sel[13], zeta_w),
          "w_3_next_eval": ev(wire_polys[3], zeta_w), "w_4_next_eval": ev(wire_polys[4], zeta_w)}
    # round 5
    terms = [(sel[j], we[j]) for j in range(4)] + [(sel[4], we[0] * we[1] % r), (sel[5], we[2] * we[3] % r)]
    terms += [(sel[6 + j], pow(we[j], 5, r)) for j in range(4)]
    terms += [(sel[12], we[0] * we[1] * we[2] * we[3] * we[4] % r), (sel[10], (-we[4]) % r), (sel[11], 1)]
    vanish = (pow(zeta, n, r) - 1) % r
    lagrange_1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r
    cf = alpha
    for j in range(W):
        cf = cf * (we[j] + beta * k[j] % r * zeta + gamma) % r
    terms.append((z_poly, (cf + alpha * alpha % r * lagrange_1) % r))
    cf = alpha * beta % r * perm_next % r
    for j in range(W - 1):
        cf = cf * (we[j] + beta * se[j] + gamma) % r
    terms.append((sig[W - 1], (-cf) % r))
    em = lambda first, ql, ds, a0, a1, a2: (first + ql * tau % r * (ds + tau * (a0 + tau * (a1 + tau * a2))) % r) % r
    mt = em(pe["range_table_eval"], pe["q_lookup_eval"], pe["table_dom_sep_eval"], pe["key_table_eval"], we[3], we[4])
    mt_next = em(pe["range_table_next_eval"], pe["q_lookup_next_eval"], pe["table_dom_sep_next_eval"], pe["key_table_next_eval"], pe["w_3_next_eval"], pe["w_4_next_eval"])
    ml = em(we[5], pe["q_lookup_eval"], pe["q_dom_sep_eval"], we[0], we[1], we[2])
    w_inv = pow(w_n, -1, r)
    lagrange_n = vanish * w_inv % r * pow(n * (zeta - w_inv) % r, -1, r) % r
    a4, a5, a6 = (pow(alpha, e, r) for e in (4, 5, 6))
    b1, zmg = (1 + beta) % r, (zeta - w_inv) % r
    g1 = gamma * b1 % r
    terms.append((pl_poly, (a4 * lagrange_1 + a5 * lagrange_n + a6 * zmg % r * b1 % r * ((gamma + ml) % r) % r * ((g1 + mt + beta * mt_next) % r)) % r))
    terms.append((h_polys[1], (-(a6 * zmg % r * pe["prod_next_eval"] % r * ((g1 + pe["h_1_eval"] + beta * pe["h_1_next_eval"]) % r))) % r))
    zeta_n2 = (vanish + 1) * zeta % r * zeta % r
    cf = 1
    for i in range(W):
        terms.append((split[i], (-vanish) * cf % r))
        cf = cf * zeta_n2 % r
    lincomb = lambda ts: cref.poly_lincomb(curve, [p for p, _ in ts], _mont(curve, r, [s for _, s in ts]), n + 3)
    lin = lincomb(terms)

    def batched(polys, point):
        bt, cf = [], 1
        for p in polys:
            bt.append((p, cf))
            cf = cf * v % r
        return cref.poly_div_linear(curve, lincomb(bt), _mont(curve, r, [point])[0])

    opening = batched([lin] + wire_polys + [sig[i] for i in range(W - 1)] + [tab[0], tab[1], h_polys[0], sel[13], tab[2], tab[3]], zeta)
    shifted = batched([z_poly, pl_poly, tab[0], tab[1], h_polys[0], h_polys[1], sel[13], wire_polys[3], wire_polys[4], tab[2]], zeta_w)
    return {"wires_comms": wires_comms, "h_comms": h_comms, "z_comm": z_comm, "prod_lookup_comm": pl_comm, "split_comms": split_comms,
            "opening": commit(opening), "shifted": commit(shifted), "wires_evals": we, "wire_sigma_evals": se, "perm_next_eval": perm_next,
            "plookup_evals": pe, "seconds": time.perf_counter() - t_start}
