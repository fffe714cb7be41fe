"""GPU: UltraPlonk (Plookup) -- SURVEY.md 8(a) rows a5 and a12 and the Plookup branches of a9, a14-a16:
the merged table / sorted vector / Plookup product builders and the whole device-resident prover against the
big-int restatement (oracle/pyref_plonk.py), through the C ABI (mzk_plonk_pk_register_ultra,
mzk_plookup_sorted_vec_dev, mzk_plookup_product_dev, mzk_plonk_quotient_ultra_dev)."""
import random

import numpy as np
import pytest

import mirror_prover as MP          # the primitive-level sequencing of the rounds: test code since round 5 (tests/mirror_prover.py)

from conftest import affine_from_limbs, build_ultra_circuit, fr_from_mont_limbs, fr_mont_limbs

pytestmark = pytest.mark.gpu

TABLES = ("range", "key", "table_dom_sep", "q_dom_sep")


def _blind(rng, r):
    return {"wires": [[rng.randrange(r), rng.randrange(r)] for _ in range(6)], "z": [rng.randrange(r) for _ in range(3)],
            "quot": [rng.randrange(r) for _ in range(5)], "h": [[rng.randrange(r) for _ in range(3)] for _ in range(2)],
            "prod_lookup": [rng.randrange(r) for _ in range(3)]}


def _make_prover(mj, c, log_n, sel, sigma_vals, k, plookup, srs_beta):
    n = 1 << log_n
    dom = mj.Radix2EvaluationDomain(c, log_n)
    sel_polys = [dom.ifft(fr_mont_limbs(c, s)) for s in sel]
    sig_polys = [dom.ifft(fr_mont_limbs(c, s)) for s in sigma_vals]
    tabs = {name: dom.ifft(fr_mont_limbs(c, plookup[key])) for name, key in zip(mj.plonk.PLOOKUP_TABLE_POLYS, TABLES)}
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)
    return MP.TurboPlonkProver(c, n, sel_polys, sig_polys, k, ck, plookup=tabs), ck


@pytest.mark.parametrize("curve_id,log_n", [(0, 5), (1, 6)])
def test_sorted_vec_and_lookup_product(gpu, mj, pyref, curve_id, log_n):
    """constraint_system.rs:1290-1417 on the device vs the restatement: merged table, merged lookup witness,
    sorted vector (first-occurrence rule with a heavily duplicated table: the range table is zero-padded),
    Plookup product coefficients."""
    import pyref_plonk as PP
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    n, r = 1 << log_n, c.r
    rng = random.Random(7 + curve_id)
    sel, sigma_vals, k, w, pi, plookup = build_ultra_circuit(pc, log_n, rng)
    prover, ck = _make_prover(mj, c, log_n, sel, sigma_vals, k, plookup, 12345)
    tau, beta, gamma = (rng.randrange(r) for _ in range(3))
    wv = np.stack([fr_mont_limbs(c, col) for col in w])
    table, lookup, sorted_vec = mj.plonk.compute_lookup_sorted_vec(prover.pk, tau, wv)
    host = lambda t: fr_from_mont_limbs(c, t.cpu().numpy().view(np.uint64))
    want_table = PP.merged_table_values(pc, tau, plookup, sel[13], w)
    want_lookup = PP.merged_lookup_values(pc, tau, plookup, sel[13], w)
    want_sorted = PP.sorted_lookup_vec(want_table, want_lookup[:n - 1])
    assert host(table) == want_table and host(lookup) == want_lookup
    assert host(sorted_vec) == want_sorted
    prod = mj.plonk.compute_lookup_prod_polynomial(prover.pk, beta, gamma, table, lookup, sorted_vec)
    want_prod = pyref.ntt_fast(pc, PP.lookup_product_values(pc, n, tau, beta, gamma, want_table, want_lookup, want_sorted), log_n, 1, inverse=True)
    assert host(prod) == want_prod
    # a value outside the table: the reference's ParameterError (constraint_system.rs:1410-1412)
    w[5][1] = 1 << 20
    with pytest.raises(mj.plonk.PlonkError, match="outside the table"):
        mj.plonk.compute_lookup_sorted_vec(prover.pk, tau, np.stack([fr_mont_limbs(c, col) for col in w]))
    # the last row is not looked up (only the first n-1 are, :1383)
    w[5][1] = 0
    w[5][n - 1] = 1 << 20
    mj.plonk.compute_lookup_sorted_vec(prover.pk, tau, np.stack([fr_mont_limbs(c, col) for col in w]))
    prover.release()
    ck.release()


@pytest.mark.parametrize("curve_id,log_n", [(0, 5), (1, 5)])
def test_ultra_prover_core_matches_bigint_restatement(gpu, mj, pyref, curve_id, log_n):
    import pyref_plonk as PP
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    n, r = 1 << log_n, c.r
    rng = random.Random(4242 + curve_id)
    sel, sigma_vals, k, w, pi, plookup = build_ultra_circuit(pc, log_n, rng)
    blind = _blind(rng, r)
    ch = {x: rng.randrange(r) for x in ("tau", "beta", "gamma", "alpha", "zeta", "v")}
    srs_beta = rng.randrange(r)
    want = PP.prove_core(pc, log_n, sel, sigma_vals, k, w, pi, blind, ch, srs_beta, plookup=plookup)
    assert want["divisible"] and want["quot_degree_ok"], "the test circuit must be satisfied"
    prover, ck = _make_prover(mj, c, log_n, sel, sigma_vals, k, plookup, srs_beta)
    proof = prover.prove(np.stack([fr_mont_limbs(c, col) for col in w]), fr_mont_limbs(c, pi), mj.prover.ProverChallenges(**ch),
                         mj.prover.Blinders(blind["wires"], blind["z"], blind["quot"], blind["h"], blind["prod_lookup"]))
    host = lambda t: fr_from_mont_limbs(c, t.cpu().numpy().view(np.uint64))
    strip = PP.pstrip
    for i in range(6):
        assert strip(host(prover.last["wire_polys"][i])) == strip(want["wire_polys"][i]), ("wire poly", i)
        assert strip(host(prover.last["split"][i])) == strip(want["split"][i]), ("split quotient", i)
    for i in range(2):
        assert strip(host(prover.last["h_polys"][i])) == strip(want["h_polys"][i]), ("h poly", i)
    assert host(prover.last["sorted_vec"]) == want["sorted_vec"]
    assert strip(host(prover.last["prod_lookup_poly"])) == strip(want["prod_lookup_poly"])
    assert strip(host(prover.last["z_poly"])) == strip(want["z_poly"])
    assert strip(host(prover.last["quot"])) == strip(want["quot"])
    assert strip(host(prover.last["lin"])) == strip(want["lin_poly"])
    assert strip(host(prover.last["opening"])) == strip(want["opening_poly"])
    assert strip(host(prover.last["shifted"])) == strip(want["shifted_opening_poly"])
    assert proof.wires_evals == want["wires_evals"] and proof.wire_sigma_evals == want["wire_sigma_evals"]
    assert proof.perm_next_eval == want["perm_next_eval"]
    assert proof.plookup_evals == want["plookup_evals"] and tuple(sorted(proof.plookup_evals)) == tuple(sorted(mj.prover.PLOOKUP_EVALS))
    G = pyref.g1_gen(pc)
    dl = want["commit_dlogs"]
    pt = lambda cm: affine_from_limbs(pc, cm.xy)
    for i in range(6):
        assert pt(proof.wires_poly_comms[i]) == pyref.g1_mul(pc, dl["wires"][i], G), ("wire commitment", i)
        assert pt(proof.split_quot_poly_comms[i]) == pyref.g1_mul(pc, dl["split"][i], G), ("quotient commitment", i)
    for i in range(2):
        assert pt(proof.h_poly_comms[i]) == pyref.g1_mul(pc, dl["h"][i], G)
    assert pt(proof.prod_lookup_poly_comm) == pyref.g1_mul(pc, dl["prod_lookup"], G)
    assert pt(proof.prod_perm_poly_comm) == pyref.g1_mul(pc, dl["z"], G)
    assert pt(proof.opening_proof) == pyref.g1_mul(pc, dl["opening"], G)
    assert pt(proof.shifted_opening_proof) == pyref.g1_mul(pc, dl["shifted_opening"], G)
    # transcript-driven run: challenges squeezed between the rounds, Plookup messages included (snark.rs:290-360)
    src = mj.prover.TranscriptChallenges(prover, [pi[3]])
    proof2 = prover.prove(np.stack([fr_mont_limbs(c, col) for col in w]), fr_mont_limbs(c, pi), src,
                          mj.prover.Blinders(blind["wires"], blind["z"], blind["quot"], blind["h"], blind["prod_lookup"]))
    ch2 = {x: src.challenges[x] for x in ("tau", "beta", "gamma", "alpha", "zeta", "v")}
    want2 = PP.prove_core(pc, log_n, sel, sigma_vals, k, w, pi, blind, ch2, srs_beta, plookup=plookup)
    assert pt(proof2.opening_proof) == pyref.g1_mul(pc, want2["commit_dlogs"]["opening"], G)
    assert pt(proof2.shifted_opening_proof) == pyref.g1_mul(pc, want2["commit_dlogs"]["shifted_opening"], G)
    assert proof2.plookup_evals == want2["plookup_evals"]
    prover.release()
    ck.release()


def test_ultra_prover_large_domain_identities(gpu, mj, pyref):
    """n = 2^12, BN254 (the curve of config C5): the properties that need no restatement -- the quotient has exactly the
    degree the reference asserts (prover.rs:916-919; anything else means the numerator was not divisible by Z_H), and
    the linearisation polynomial at zeta cancels the verifier's constant term (verifier.rs:340-414)."""
    import pyref_plonk as PP
    curve_id, log_n = 1, 12
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    n, r = 1 << log_n, c.r
    rng = random.Random(99)
    sel, sigma_vals, k, w, pi, plookup = build_ultra_circuit(pc, log_n, rng, range_bits=8)
    blind = _blind(rng, r)
    ch = {x: rng.randrange(r) for x in ("tau", "beta", "gamma", "alpha", "zeta", "v")}
    prover, ck = _make_prover(mj, c, log_n, sel, sigma_vals, k, plookup, rng.randrange(r))
    proof = prover.prove(np.stack([fr_mont_limbs(c, col) for col in w]), fr_mont_limbs(c, pi), mj.prover.ProverChallenges(**ch),
                         mj.prover.Blinders(blind["wires"], blind["z"], blind["quot"], blind["h"], blind["prod_lookup"]))
    quot = prover.last["quot"].cpu().numpy().view(np.uint64)
    deg = 6 * (n + 1) + 2
    assert quot[deg].any() and not quot[deg + 1:].any()
    a, b, g, zeta = ch["alpha"], ch["beta"], ch["gamma"], ch["zeta"]
    w_inv = pow(pc.root_of_unity(log_n), -1, r)
    vanish = (pow(zeta, n, r) - 1) % r
    l1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r
    ln = vanish * w_inv % r * pow(n * (zeta - w_inv) % r, -1, r) % r
    pi_at_zeta = l1 * 0                                                   # PI(zeta) = sum_i pi_i L_i(zeta); one non-zero entry at row 3
    w_n = pc.root_of_unity(log_n)
    x3 = pow(w_n, 3, r)
    pi_at_zeta = pi[3] * (vanish * x3 % r * pow(n * (zeta - x3) % r, -1, r) % r) % r
    we, se, e = proof.wires_evals, proof.wire_sigma_evals, proof.plookup_evals
    tmp = (pi_at_zeta - a * a * l1) % r
    acc = a * proof.perm_next_eval % r * (g + we[-1]) % r
    for w_e, s_e in zip(we[:-1], se):
        acc = acc * (g + w_e + b * s_e) % r
    g1 = g * (1 + b) % r
    pc_ = (ln * (e["h_1_eval"] - e["h_2_next_eval"] - a * a) - a * l1
           - a ** 3 * (zeta - w_inv) % r * e["prod_next_eval"] % r * (g1 + e["h_1_eval"] + b * e["h_1_next_eval"]) % r * (g1 + b * e["h_2_next_eval"])) % r
    const = (tmp - acc + a ** 3 * pc_) % r
    lin_at_zeta = mj.poly.evaluate(c, prover.last["lin"], zeta)[0]
    assert (lin_at_zeta + const) % r == 0
    prover.release()
    ck.release()


def test_ultra_quotient_kernels_limb_extremes(gpu, mj, cref):
    """Both UltraPlonk quotient kernels (plonk.cuh, reduced-radix lazy arithmetic) at their worst-case limb patterns, against a
    big-int evaluation of the closure of prover.rs:605-659 with compute_quotient_plookup_contribution (:773-888) at every point
    of the quotient coset.  All 35 operand streams are chosen as EVALUATIONS (full-length polynomials interpolate them)."""
    import torch
    from test_plonk_gpu import _adversarial_field_values
    for curve_id in (0, 1):
        c = mj.params.CURVES[curve_id]
        r = c.r
        log_n, n = 2, 4
        m = 8 * n
        rng = random.Random(8192 + curve_id)
        gmont = mj.params.fr_to_mont(c, [c.fr_generator])[0]
        E = [_adversarial_field_values(c, rng, m) for _ in range(35)]       # 14 sel, 6 sig, 4 tab, 6 wires, z, pi, h1, h2, pl
        to_poly = lambda vals: cref.ntt(curve_id, fr_mont_limbs(c, vals), log_n + 3, True, gmont, threads=1)
        polys = [to_poly(v) for v in E]
        k = _adversarial_field_values(c, rng, 6)
        tau, alpha, beta, gamma = _adversarial_field_values(c, rng, 4)
        tabs = {name: polys[20 + i] for i, name in enumerate(mj.plonk.PLOOKUP_TABLE_POLYS)}
        pk = mj.plonk.ProvingKeyDevice.register(c, n, polys[:14], polys[14:20], k, plookup=tabs)
        slab = torch.from_numpy(np.stack(polys[24:35]).view(np.int64)).cuda().contiguous()
        out = torch.empty((m, 4), dtype=torch.int64, device="cuda")
        mj.plonk.compute_quotient_polynomial_dev(pk, mj.plonk.Challenges(alpha, beta, gamma, tau), slab, m, out)
        got = out.cpu().numpy().view(np.uint64)
        # ---- the closure, point by point
        sel, sig, tab, w = E[:14], E[14:20], E[20:24], E[24:30]
        z, pi, h1, h2, pl = E[30], E[31], E[32], E[33], E[34]
        w_m = pow(c.fr_generator, (r - 1) // m, r)
        w_n_inv = pow(pow(w_m, 8, r), -1, r)
        zh_inv = [pow((pow(c.fr_generator * pow(w_m, i, r) % r, n, r) - 1) % r, -1, r) for i in range(8)]
        b1, g1 = (1 + beta) % r, gamma * (1 + beta) % r
        merged = lambda first, q, ds, a0, a1, a2: (first + q * tau % r * (ds + tau * (a0 + tau * (a1 + tau * a2))) % r) % r
        t_evals = []
        for i in range(m):
            nx = (i + 8) % m
            x = c.fr_generator * pow(w_m, i, r) % r
            wi = [w[j][i] for j in range(6)]
            t = (sel[11][i] + pi[i] + sum(sel[j][i] * wi[j] for j in range(4)) + sel[4][i] * wi[0] * wi[1] + sel[5][i] * wi[2] * wi[3]
                 + sel[12][i] * wi[0] * wi[1] * wi[2] * wi[3] * wi[4] + sum(sel[6 + j][i] * pow(wi[j], 5, r) for j in range(4)) - sel[10][i] * wi[4]) % r
            acc1, acc2 = z[i], z[nx]
            for j in range(6):
                acc1 = acc1 * (wi[j] + gamma + k[j] * x % r * beta) % r
                acc2 = acc2 * (wi[j] + gamma + sig[j][i] * beta) % r
            lag_1 = pow(n * (x - 1) % r, -1, r)
            lag_n = w_n_inv * pow(n * (x - w_n_inv) % r, -1, r) % r
            t1 = (t + alpha * (acc1 - acc2)) % r
            t2 = alpha * alpha % r * (z[i] - 1) % r * lag_1 % r
            mt = merged(tab[0][i], sel[13][i], tab[2][i], tab[1][i], wi[3], wi[4])
            mt_next = merged(tab[0][nx], sel[13][nx], tab[2][nx], tab[1][nx], w[3][nx], w[4][nx])
            ml = merged(wi[5], sel[13][i], tab[3][i], wi[0], wi[1], wi[2])
            a3 = pow(alpha, 3, r)
            t2 = (t2 + a3 * (h1[i] - h2[nx]) % r * lag_n + a3 * alpha % r * (pl[i] - 1) % r * lag_1 + a3 * alpha * alpha % r * (pl[i] - 1) % r * lag_n) % r
            term3 = (x - w_n_inv) * (pl[i] * b1 % r * (gamma + ml) % r * (g1 + mt + beta * mt_next) % r
                                     - pl[nx] * (g1 + h1[i] + beta * h1[nx]) % r * (g1 + h2[i] + beta * h2[nx]) % r) % r
            t1 = (t1 + pow(alpha, 6, r) * term3) % r
            t_evals.append((t1 * zh_inv[i % 8] + t2) % r)
        want = cref.ntt(curve_id, fr_mont_limbs(c, t_evals), log_n + 3, True, gmont, threads=1)
        assert np.array_equal(got, want), curve_id
        pk.release()
