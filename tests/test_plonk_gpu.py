"""GPU parity for the device-resident TurboPlonk quotient round: against the C
restatement of prover.rs:512-759 on random inputs, and -- independent of any oracle -- the degree and
divisibility properties the reference's own tests check (snark.rs:1282-1408) on a satisfied circuit."""
import random

import numpy as np
import pytest

from conftest import fr_from_mont_limbs, fr_mont_limbs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("log_n", [1, 3, 6, 9])
def test_quotient_matches_c_oracle(gpu, mj, cref, curve_id, log_n):
    c = mj.params.CURVES[curve_id]
    n = 1 << log_n
    rng = random.Random(log_n * 7 + curve_id)
    # wires carry one blinding coefficient pair (n+2), z three (n+3): prover.rs:79-83,133-138
    polys = mj.params.random_fr_mont(c, 25 * (n + 3), seed=log_n).reshape(25, n + 3, 4)
    polys[:18, n:] = 0          # selectors and sigmas have degree < n
    polys[18:23, n + 2:] = 0
    k = [rng.randrange(1, c.r) for _ in range(5)]
    ch = mj.plonk.Challenges(rng.randrange(c.r), rng.randrange(c.r), rng.randrange(c.r))
    pk = mj.plonk.ProvingKeyDevice.register(c, n, list(polys[:13, :n]), list(polys[13:18, :n]), k)
    got = mj.plonk.compute_quotient_polynomial(pk, ch, list(polys[18:23, :n + 2]), polys[23], polys[24, :n])
    pi_padded = polys.copy()
    pi_padded[24, n:] = 0
    want = cref.plonk_quotient(curve_id, log_n, pi_padded, mj.params.fr_to_mont(c, k), *mj.params.fr_to_mont(c, [ch.alpha, ch.beta, ch.gamma]),
                               threads=8)
    assert np.array_equal(got, want)
    # a second proof against the same resident key (different witness polynomials and challenges)
    polys2 = polys.copy()
    polys2[18:] = mj.params.random_fr_mont(c, 7 * (n + 3), seed=99).reshape(7, n + 3, 4)
    ch2 = mj.plonk.Challenges(5, 7, 11)
    got2 = mj.plonk.compute_quotient_polynomial(pk, ch2, list(polys2[18:23]), polys2[23], polys2[24])
    want2 = cref.plonk_quotient(curve_id, log_n, polys2, mj.params.fr_to_mont(c, k), *mj.params.fr_to_mont(c, [5, 7, 11]), threads=8)
    assert np.array_equal(got2, want2)
    pk.release()


def _interpolate(mj, c, log_n, values):
    return mj.Radix2EvaluationDomain(c, log_n).ifft(fr_mont_limbs(c, values))


@pytest.mark.parametrize("curve_id", [0, 1])
def test_quotient_of_a_satisfied_circuit_is_a_low_degree_polynomial(gpu, mj, cref, pyref, curve_id):
    """Add / mul / x^5 / constant gates with the identity permutation (z = 1, sigma_j = k_j X): the gate
    polynomial vanishes on H, so t = gate / Z_H has degree < 5n and t * Z_H == gate at a random point;
    breaking one witness value makes the high coefficients non-zero."""
    c = mj.params.CURVES[curve_id]
    pc = pyref.CURVES[curve_id]
    log_n, n = 6, 64
    rng = random.Random(11 + curve_id)
    r = c.r
    w = [[rng.randrange(r) for _ in range(n)] for _ in range(5)]
    sel = [[0] * n for _ in range(13)]
    for i in range(n):
        kind = i % 4
        if kind == 0:      # w0 + w1 = w4
            sel[0][i] = sel[1][i] = 1; sel[10][i] = 1
            w[4][i] = (w[0][i] + w[1][i]) % r
        elif kind == 1:    # 3*w0*w1 + w2*w3 = w4
            sel[4][i] = 3; sel[5][i] = 1; sel[10][i] = 1
            w[4][i] = (3 * w[0][i] * w[1][i] + w[2][i] * w[3][i]) % r
        elif kind == 2:    # w0^5 + 2*w3^5 + w0*w1*w2*w3*w4' ... keep the ecc gate separate
            sel[6][i] = 1; sel[9][i] = 2; sel[10][i] = 1
            w[4][i] = (pow(w[0][i], 5, r) + 2 * pow(w[3][i], 5, r)) % r
        else:              # q_ecc * w0 w1 w2 w3 w4 + q_c = 0 with w4 chosen, and a linear term on w2
            sel[12][i] = 1
            w[4][i] = rng.randrange(r)
            prod = w[0][i] * w[1][i] % r * w[2][i] % r * w[3][i] % r * w[4][i] % r
            sel[11][i] = (-prod) % r
    k = [1, 7, 13, 17, 23]
    sel_polys = [_interpolate(mj, c, log_n, s) for s in sel]
    sigma_polys = [fr_mont_limbs(c, [0, kj] + [0] * (n - 2)) for kj in k]          # sigma_j(X) = k_j X
    z_poly = fr_mont_limbs(c, [1])
    pi_poly = fr_mont_limbs(c, [0])
    pk = mj.plonk.ProvingKeyDevice.register(c, n, sel_polys, sigma_polys, k)
    ch = mj.plonk.Challenges(rng.randrange(r), rng.randrange(r), rng.randrange(r))
    wire_polys = [_interpolate(mj, c, log_n, col) for col in w]
    t = mj.plonk.compute_quotient_polynomial(pk, ch, wire_polys, z_poly, pi_poly)
    assert not t[5 * n:].any(), "quotient of a satisfied circuit must have degree < 5n"
    # t(x) * Z_H(x) == gate(x) at a random point (big-int evaluation of every polynomial)
    x = rng.randrange(r)
    ev = lambda poly: pyref.poly_eval(pc, fr_from_mont_limbs(c, poly), x)
    W = [ev(p) for p in wire_polys]
    S = [ev(p) for p in sel_polys]
    gate = (S[11] + sum(S[j] * W[j] for j in range(4)) + S[4] * W[0] * W[1] + S[5] * W[2] * W[3]
            + S[12] * W[0] * W[1] * W[2] * W[3] * W[4] + sum(S[6 + j] * pow(W[j], 5, r) for j in range(4)) - S[10] * W[4]) % r
    assert ev(t) * (pow(x, n, r) - 1) % r == gate
    # the host-pointer entry point over a key that holds only the 6 needed residue classes (the hosts' default) returns the same
    # 8n coefficients as over the whole-domain key; over 5 classes -- which cannot determine a degree-(5n + 7) quotient -- it refuses
    pk6 = mj.plonk.ProvingKeyDevice.register(c, n, sel_polys, sigma_polys, k, classes=mj.plonk.quotient_classes_needed(5, n, top=False))
    assert np.array_equal(mj.plonk.compute_quotient_polynomial(pk6, ch, wire_polys, z_poly, pi_poly), t)
    pk5 = mj.plonk.ProvingKeyDevice.register(c, n, sel_polys, sigma_polys, k, classes=[0, 1, 2, 3, 4])
    with pytest.raises(Exception):
        mj.plonk.compute_quotient_polynomial(pk5, ch, wire_polys, z_poly, pi_poly)
    pk5.release()
    # an unsatisfied gate breaks divisibility
    w[4][5] = (w[4][5] + 1) % r
    bad_polys = [_interpolate(mj, c, log_n, col) for col in w]
    bad = mj.plonk.compute_quotient_polynomial(pk, ch, bad_polys, z_poly, pi_poly)
    assert bad[5 * n:].any()
    bad6 = mj.plonk.compute_quotient_polynomial(pk6, ch, bad_polys, z_poly, pi_poly)
    assert bad6[5 * n + 8:6 * n].any() and not bad6[6 * n:].any(), "from 6 classes the interpolant has degree < 6n, and not 5n + 7"
    pk.release()
    pk6.release()


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("log_n", [3, 4, 6, 9, 11])             # (11: every class in one launch per step -- plonk.hip quotient_chunked_run, from 2^10 gates on)
def test_quotient_from_one_class_fewer_and_the_top_coefficients(gpu, mj, pyref, curve_id, log_n):
    """mzk_plonk_quotient_top_dev + mzk_plonk_quotient_combine_top_dev: W classes and the W + 3 top coefficients of the numerator give the
    same 8n coefficients as the whole-domain key.  The circuit of the test above (add / mul / x^5 / ecc gates, identity permutation
    sigma_j = k_j X) with BLINDED polynomials: w_j + (b0 + b1 X) Z_H and z = 1 + (c0 + c1 X + c2 X^2) Z_H keep the numerator divisible
    by Z_H and bring the quotient to its full degree 5 (n + 1) + 2 with the q_hash / q_ecc terms inside the top window."""
    import torch
    c = mj.params.CURVES[curve_id]
    pc = pyref.CURVES[curve_id]
    n = 1 << log_n
    rng = random.Random(1234 + 7 * curve_id + log_n)
    r = c.r
    w = [[rng.randrange(r) for _ in range(n)] for _ in range(5)]
    sel = [[0] * n for _ in range(13)]
    for i in range(n):
        kind = i % 4
        if kind == 0:
            sel[0][i] = sel[1][i] = 1; sel[10][i] = 1
            w[4][i] = (w[0][i] + w[1][i]) % r
        elif kind == 1:
            sel[4][i] = 3; sel[5][i] = 1; sel[10][i] = 1
            w[4][i] = (3 * w[0][i] * w[1][i] + w[2][i] * w[3][i]) % r
        elif kind == 2:
            sel[6][i] = 1; sel[7][i] = 5; sel[8][i] = rng.randrange(r); sel[9][i] = 2; sel[10][i] = 1
            w[4][i] = (pow(w[0][i], 5, r) + 5 * pow(w[1][i], 5, r) + sel[8][i] * pow(w[2][i], 5, r) + 2 * pow(w[3][i], 5, r)) % r
        else:
            sel[12][i] = rng.randrange(1, r)
            w[4][i] = rng.randrange(r)
            prod = w[0][i] * w[1][i] % r * w[2][i] % r * w[3][i] % r * w[4][i] % r
            sel[11][i] = (-sel[12][i] * prod) % r
    k = [1, 7, 13, 17, 23]
    sel_polys = [_interpolate(mj, c, log_n, s_) for s_ in sel]
    sigma_polys = [fr_mont_limbs(c, [0, kj] + [0] * (n - 2)) for kj in k]

    def blinded(coeffs, blind):                                   # p + blind(X) (X^n - 1)
        out = list(coeffs) + [0] * (n + len(blind) - len(coeffs))
        for i, b in enumerate(blind):
            out[n + i] = (out[n + i] + b) % r
            out[i] = (out[i] - b) % r
        return out
    wire_int = [blinded(fr_from_mont_limbs(c, _interpolate(mj, c, log_n, col)), [rng.randrange(1, r), rng.randrange(1, r)]) for col in w]
    z_int = blinded([1] + [0] * (n - 1), [rng.randrange(1, r) for _ in range(3)])
    rows = np.zeros((7, n + 3, 4), dtype=np.uint64)
    for j in range(5):
        rows[j, :n + 2] = fr_mont_limbs(c, wire_int[j])
    rows[5] = fr_mont_limbs(c, z_int)
    ch = mj.plonk.Challenges(rng.randrange(r), rng.randrange(r), rng.randrange(r))
    pk = mj.plonk.ProvingKeyDevice.register(c, n, sel_polys, sigma_polys, k)
    want = mj.plonk.compute_quotient_polynomial(pk, ch, [rows[j] for j in range(5)], rows[5], rows[6])
    assert want[5 * n + 7].any() and not want[5 * n + 8:].any(), "the blinded quotient has degree exactly 5 (n + 1) + 2"
    classes = mj.plonk.quotient_classes_needed(5, n)
    assert classes == [0, 1, 2, 3, 4] and mj.plonk.quotient_classes_needed(5, n, top=False) == ([0, 1, 2, 3, 4, 5] if n > 8 else list(range(8)))
    for cl in (classes, [1, 3, 4, 6, 7]):
        pk5 = mj.plonk.ProvingKeyDevice.register(c, n, sel_polys, sigma_polys, k, classes=cl)
        slab = torch.from_numpy(rows.view(np.int64)).cuda()
        rem = mj.plonk.compute_quotient_chunked_dev(pk5, ch, slab, n + 3)
        top = mj.plonk.compute_quotient_top_dev(pk5, ch, slab, n + 3)
        got = mj.plonk.combine_quotient_classes(c, n, rem, classes=cl, top=top, n_top=8).cpu().numpy().view(np.uint64)
        assert np.array_equal(top.cpu().numpy().view(np.uint64)[:8], want[5 * n:5 * n + 8]), "top coefficients"
        assert np.array_equal(got, want)
        pk5.release()
    pk.release()


def test_quotient_device_resident_and_errors(gpu, mj, cref):
    import torch
    c = mj.params.BLS12_381
    log_n, n = 8, 256
    m = 8 * n
    polys = mj.params.random_fr_mont(c, 25 * n, seed=4).reshape(25, n, 4)
    k = [1, 2, 3, 4, 5]
    pk = mj.plonk.ProvingKeyDevice.register(c, n, list(polys[:13]), list(polys[13:18]), k)
    slab = np.zeros((7, m, 4), dtype=np.uint64)
    slab[:, :n] = polys[18:]
    d = torch.from_numpy(slab.view(np.int64)).cuda()
    out = torch.empty((m, 4), dtype=torch.int64, device="cuda")
    ch = mj.plonk.Challenges(3, 5, 9)
    mj.plonk.compute_quotient_polynomial_dev(pk, ch, d, n, out)
    torch.cuda.synchronize()
    want = cref.plonk_quotient(0, log_n, polys, mj.params.fr_to_mont(c, k), *mj.params.fr_to_mont(c, [3, 5, 9]), threads=8)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    with pytest.raises(mj.plonk.PlonkError):
        mj.plonk.compute_quotient_polynomial(pk, ch, list(polys[18:22]), polys[23], polys[24])       # 4 wire polys
    with pytest.raises(mj.plonk.PlonkError):
        mj.plonk.ProvingKeyDevice.register(c, n, list(polys[:12]), list(polys[13:18]), k)
    pk.release()
    with pytest.raises(mj.MzkError):
        mj.plonk.compute_quotient_polynomial(pk.__class__(c, 424242, n), ch, list(polys[18:23]), polys[23], polys[24])


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("log_n", [1, 4, 11, 14])
def test_perm_product_matches_c_oracle(gpu, mj, cref, curve_id, log_n):
    """Round 2 (N2): z = ifft(running product) against the restatement of constraint_system.rs:1197-1223
    (one field division per gate there; shared inversions + a parallel prefix product here)."""
    c = mj.params.CURVES[curve_id]
    n = 1 << log_n
    rng = random.Random(log_n + 100 * curve_id)
    fixed = mj.params.random_fr_mont(c, 18 * n, seed=log_n + 5).reshape(18, n, 4)
    k = [rng.randrange(1, c.r) for _ in range(5)]
    pk = mj.plonk.ProvingKeyDevice.register(c, n, list(fixed[:13]), list(fixed[13:]), k)
    wires = mj.params.random_fr_mont(c, 5 * n, seed=log_n + 6).reshape(5, n, 4)
    beta, gamma = rng.randrange(c.r), rng.randrange(c.r)
    got = mj.plonk.compute_prod_permutation_polynomial(pk, beta, gamma, wires)
    sigma_vals = np.stack([cref.ntt(curve_id, fixed[13 + i], log_n, False, None, threads=4) for i in range(5)])
    want = cref.plonk_perm_product(curve_id, log_n, wires, sigma_vals, mj.params.fr_to_mont(c, k), *mj.params.fr_to_mont(c, [beta, gamma]), threads=4)
    assert np.array_equal(got, want)
    pk.release()


def test_perm_product_of_a_valid_permutation(gpu, mj, pyref):
    """With sigma a true permutation of the extended identity and a witness that respects it, the grand
    product telescopes: z(w^j) as computed, times the last row's ratio, returns to 1; the identity
    permutation gives z = 1 exactly."""
    c = mj.params.BLS12_381
    pc = pyref.BLS12_381
    log_n, n = 5, 32
    r = c.r
    rng = random.Random(3)
    k = [1, 7, 13, 17, 23]
    w_n = pc.root_of_unity(log_n)
    ident = [[k[i] * pow(w_n, j, r) % r for j in range(n)] for i in range(5)]
    # a permutation made of 3-cycles over random cells; cells of one cycle carry one witness value
    cells = [(i, j) for i in range(5) for j in range(n)]
    rng.shuffle(cells)
    perm = {cell: cell for cell in cells}
    wires = [[rng.randrange(r) for _ in range(n)] for _ in range(5)]
    for q in range(0, 60, 3):
        a, b, d = cells[q], cells[q + 1], cells[q + 2]
        perm[a], perm[b], perm[d] = b, d, a
        v = rng.randrange(r)
        for (i, j) in (a, b, d):
            wires[i][j] = v
    sigma_vals = [[ident[perm[(i, j)][0]][perm[(i, j)][1]] for j in range(n)] for i in range(5)]
    dom = mj.Radix2EvaluationDomain(c, log_n)
    sigma_polys = [dom.ifft(fr_mont_limbs(c, sv)) for sv in sigma_vals]
    zero_sel = [np.zeros((1, 4), dtype=np.uint64)] * 13
    pk = mj.plonk.ProvingKeyDevice.register(c, n, zero_sel, sigma_polys, k)
    beta, gamma = rng.randrange(r), rng.randrange(r)
    z_poly = mj.plonk.compute_prod_permutation_polynomial(pk, beta, gamma, np.stack([fr_mont_limbs(c, col) for col in wires]))
    z_vals = fr_from_mont_limbs(c, dom.fft(z_poly))
    assert z_vals[0] == 1
    num = den = 1
    for i in range(5):
        num = num * (wires[i][n - 1] + gamma + beta * ident[i][n - 1]) % r
        den = den * (wires[i][n - 1] + gamma + beta * sigma_vals[i][n - 1]) % r
    assert z_vals[n - 1] * num % r == den % r * 1, "grand product of a satisfied permutation must close to 1"
    # a broken copy constraint does not close
    wires[cells[0][0]][cells[0][1]] = (wires[cells[0][0]][cells[0][1]] + 1) % r
    z_bad = fr_from_mont_limbs(c, dom.fft(mj.plonk.compute_prod_permutation_polynomial(pk, beta, gamma, np.stack([fr_mont_limbs(c, col) for col in wires]))))
    num = den = 1
    for i in range(5):
        num = num * (wires[i][n - 1] + gamma + beta * ident[i][n - 1]) % r
        den = den * (wires[i][n - 1] + gamma + beta * sigma_vals[i][n - 1]) % r
    assert z_bad[n - 1] * num % r != den % r
    pk.release()
    # identity permutation: z == 1
    id_polys = [fr_mont_limbs(c, [0, kj] + [0] * (n - 2)) for kj in k]
    pk = mj.plonk.ProvingKeyDevice.register(c, n, zero_sel, id_polys, k)
    z1 = mj.plonk.compute_prod_permutation_polynomial(pk, beta, gamma, mj.params.random_fr_mont(c, 5 * n, seed=1).reshape(5, n, 4))
    assert fr_from_mont_limbs(c, z1) == [1] + [0] * (n - 1)
    pk.release()


def _adversarial_field_values(c, rng, count):
    """Field elements x whose INTERNAL image x * 2^261 mod r (what the reduced-radix quotient kernels hold, plonk.cuh) has extreme
    29-bit limbs: r - 1, all eight low limbs at 2^29 - 1 under the largest admissible top limb, 0, 1, and a few random ones."""
    r = c.r
    rp_inv = pow(1 << 261, -1, r)
    low = (1 << 232) - 1
    extremes = [r - 1, low + (((r >> 232) - 1) << 232), low, 0, 1, (1 << 232), r - 2]
    return [(extremes[rng.randrange(len(extremes))] if rng.random() < 0.8 else rng.randrange(r)) * rp_inv % r for _ in range(count)]


@pytest.mark.parametrize("curve_id", [0, 1])
def test_quotient_kernel_limb_extremes(gpu, mj, cref, curve_id):
    """The lazy-reduction bounds of the quotient kernel (plonk.cuh) at their worst case: every operand stream -- selector, sigma,
    wire, z and public-input EVALUATIONS on the quotient coset, and the challenges -- is drawn from values whose internal limbs
    are all-ones / modulus-minus-one patterns.  Polynomials of full length 8n interpolate those evaluations."""
    c = mj.params.CURVES[curve_id]
    log_n, n = 3, 8
    m = 8 * n
    rng = random.Random(4096 + curve_id)
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    evals = fr_mont_limbs(c, _adversarial_field_values(c, rng, 25 * m)).reshape(25, m, 4)
    polys = np.stack([cref.ntt(curve_id, evals[i], log_n + 3, True, g, threads=1) for i in range(25)])       # coset iFFT: coefficients
    k = _adversarial_field_values(c, rng, 5)
    a, b, gm = _adversarial_field_values(c, rng, 3)
    pk = mj.plonk.ProvingKeyDevice.register(c, n, list(polys[:13]), list(polys[13:18]), k)
    got = mj.plonk.compute_quotient_polynomial(pk, mj.plonk.Challenges(a, b, gm), list(polys[18:23]), polys[23], polys[24])
    want = cref.plonk_quotient(curve_id, log_n, polys, mj.params.fr_to_mont(c, k), *mj.params.fr_to_mont(c, [a, b, gm]), threads=2)
    assert np.array_equal(got, want)
    pk.release()
