"""oracle/pyref_plonk.py -- TEST INFRASTRUCTURE ONLY.

Big-integer restatement of the arithmetic of one TurboPlonk proof (single instance, no Plookup), for
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


def prove_core(c, log_n, selector_vals, sigma_vals, k, wire_vals, pi_vals, blind, ch, srs_beta=None):
    """selector_vals: 13 x n, sigma_vals / wire_vals: 5 x n evaluations on H; pi_vals: n.
    blind: {"wires": 5 x [b0,b1], "z": [b0,b1,b2], "quot": [4 scalars]}.  ch: dict beta gamma alpha zeta v.
    Returns every polynomial, the 10 evaluations and (if srs_beta) the discrete logs of the 13 commitments."""
    r = c.r
    n = 1 << log_n
    w_n = c.root_of_unity(log_n)
    intt = lambda vals: P.ntt_fast(c, list(vals), log_n, 1, inverse=True)
    sel = [intt(v) for v in selector_vals]
    sig = [intt(v) for v in sigma_vals]
    beta, gamma, alpha, zeta, v = (ch[x] for x in ("beta", "gamma", "alpha", "zeta", "v"))
    # round 1
    wire_polys = [mask(c, intt(wire_vals[i]), blind["wires"][i], n) for i in range(5)]
    pi_poly = intt(pi_vals)
    # round 2 (constraint_system.rs:1197-1223)
    prod = [1]
    for j in range(n - 1):
        a = b = 1
        for i in range(5):
            t = (wire_vals[i][j] + gamma) % r
            a = a * (t + beta * k[i] * pow(w_n, j, r)) % r
            b = b * (t + beta * sigma_vals[i][j]) % r
        prod.append(prod[-1] * a % r * pow(b, -1, r) % r)
    z_unmasked = intt(prod)
    z_poly = mask(c, z_unmasked, blind["z"], n)
    # round 3: t = [gate + alpha*(z prod(w + beta k X + gamma) - z(wX) prod(w + beta sigma + gamma)) + alpha^2 (z - 1) L1] / Z_H
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
    acc2 = [cf * pow(w_n, i, r) % r for i, cf in enumerate(z_poly)]          # z(w X)
    for j in range(5):
        acc1 = pmul(c, acc1, padd(c, wire_polys[j], [gamma, beta * k[j] % r]))
        acc2 = pmul(c, acc2, padd(c, padd(c, wire_polys[j], [gamma]), pscale(c, sig[j], beta)))
    perm = pscale(c, padd(c, acc1, pscale(c, acc2, r - 1)), alpha)
    l1 = intt([1] + [0] * (n - 1))                                            # L_1: 1 at w^0
    bound = pscale(c, pmul(c, padd(c, z_poly, [r - 1]), l1), alpha * alpha % r)
    numer = padd(c, padd(c, gate, perm), bound)
    quot, rem = div_by_vanishing(c, numer, n)
    quot = pstrip(quot)
    divisible = not any(rem)
    # split (prover.rs:902-960)
    expected_degree = 5 * (n + 1) + 2
    split = [quot[i * (n + 2):(i + 1) * (n + 2)] if i < 4 else quot[4 * (n + 2):] for i in range(5)]
    last = 0
    for i in range(4):
        now = blind["quot"][i]
        split[i] = list(split[i]) + [0] * (n + 2 - len(split[i]))
        split[i][0] = (split[i][0] - last) % r
        split[i].append(now)
        last = now
    if split[4]:
        split[4] = list(split[4])
        split[4][0] = (split[4][0] - last) % r
    # round 4
    ev = lambda poly, x: P.poly_eval(c, poly, x)
    wires_evals = [ev(p, zeta) for p in wire_polys]
    wire_sigma_evals = [ev(sig[i], zeta) for i in range(4)]
    perm_next_eval = ev(z_poly, zeta * w_n % r)
    # round 5: linearisation polynomial (prover.rs:963-1035, 343-358)
    we = wires_evals
    terms = [(sel[j], we[j]) for j in range(4)]
    terms += [(sel[4], we[0] * we[1] % r), (sel[5], we[2] * we[3] % r)]
    terms += [(sel[6 + j], pow(we[j], 5, r)) for j in range(4)]
    terms += [(sel[12], we[0] * we[1] % r * we[2] % r * we[3] % r * we[4] % r), (sel[10], (-we[4]) % r), (sel[11], 1)]
    vanish = (pow(zeta, n, r) - 1) % r
    lagrange_1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r
    coeff = alpha
    for j in range(5):
        coeff = coeff * (we[j] + beta * k[j] % r * zeta + gamma) % r
    coeff = (coeff + alpha * alpha % r * lagrange_1) % r
    terms.append((z_poly, coeff))
    coeff = alpha * beta % r * perm_next_eval % r
    for j in range(4):
        coeff = coeff * (we[j] + beta * wire_sigma_evals[j] + gamma) % r
    terms.append((sig[4], (-coeff) % r))
    zeta_n2 = (vanish + 1) * zeta % r * zeta % r
    cf = 1
    for i in range(5):
        terms.append((split[i], (-vanish) * cf % r))
        cf = cf * zeta_n2 % r
    lin = [0]
    for poly, s in terms:
        lin = padd(c, lin, pscale(c, poly, s))
    # opening proofs (prover.rs:362-419, 490-509)
    batch = [0]
    cf = 1
    for poly in [lin] + wire_polys + sig[:4]:
        batch = padd(c, batch, pscale(c, poly, cf))
        cf = cf * v % r
    opening = div_by_linear(c, batch, zeta)
    shifted = div_by_linear(c, z_poly, zeta * w_n % r)
    out = {"wire_polys": wire_polys, "pi_poly": pi_poly, "z_poly": z_poly, "quot": quot, "divisible": divisible,
           "quot_degree_ok": len(quot) - 1 == expected_degree, "split": split,
           "wires_evals": wires_evals, "wire_sigma_evals": wire_sigma_evals, "perm_next_eval": perm_next_eval,
           "lin_poly": lin, "opening_poly": opening, "shifted_opening_poly": shifted, "selectors": sel, "sigmas": sig}
    if srs_beta is not None:
        dl = lambda poly: P.poly_eval(c, poly, srs_beta)
        out["commit_dlogs"] = {"wires": [dl(p) for p in wire_polys], "z": dl(z_poly), "split": [dl(p) for p in split],
                               "opening": dl(opening), "shifted_opening": dl(shifted)}
    return out
