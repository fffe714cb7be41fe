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
    ntt = lambda a, inverse, coset=None, lg=log_n: cref.ntt(curve, a, lg, inverse, coset, threads=threads)
    k_m = _mont(curve, r, k)
    beta, gamma, alpha, zeta, v = (ch[x] for x in ("beta", "gamma", "alpha", "zeta", "v"))
    bm, gm, am = (_mont(curve, r, [x])[0] for x in (beta, gamma, alpha))
    sel = np.stack([ntt(selector_vals[i], True) for i in range(13)])
    sig = np.stack([ntt(sigma_vals[i], True) for i in range(W)])

    def mask(poly, b):                                                   # prover.rs:463-486
        out = np.zeros((n + len(b), 4), dtype=np.uint64)
        out[:n] = poly
        head = _ints(curve, out[:len(b)])
        out[:len(b)] = _mont(curve, r, [(h - x) % r for h, x in zip(head, b)])
        out[n:] = _mont(curve, r, b)
        return out

    commit = lambda p: cref.jac_to_affine(curve, cref.msm(curve, srs_xy[:p.shape[0]], p, scalars_are_mont=True, threads=threads))[0]
    # round 1
    wire_polys = [mask(ntt(wire_vals[i], True), blind["wires"][i]) for i in range(W)]
    pi_poly = ntt(pi_vals, True)
    wires_comms = [commit(p) for p in wire_polys]
    # round 2 (constraint_system.rs:1197-1223)
    z_poly = mask(cref.plonk_perm_product(curve, log_n, np.stack(wire_vals), np.stack(sigma_vals), k_m, bm, gm, threads=threads), blind["z"])
    z_comm = commit(z_poly)
    # round 3 (prover.rs:512-673, 902-960)
    slab = np.zeros((25, n + 3, 4), dtype=np.uint64)
    slab[:13, :n] = sel
    slab[13:18, :n] = sig
    for i in range(W):
        slab[18 + i, :n + 2] = wire_polys[i]
    slab[23] = z_poly
    slab[24, :n] = pi_poly
    quot = cref.plonk_quotient(curve, log_n, slab, k_m, am, bm, gm, threads=threads)
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
            "wires_evals": we, "wire_sigma_evals": se, "perm_next_eval": perm_next, "seconds": time.perf_counter() - t_start}
