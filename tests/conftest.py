import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def pyref():
    import pyref as P
    return P


@pytest.fixture(scope="session")
def cref():
    """The C oracle (oracle/cpu_ref.c), built on demand."""
    import cref as R
    R.lib()
    return R


@pytest.fixture(scope="session")
def mj():
    import mpc_jellyfish_amd as m
    return m


@pytest.fixture(scope="session")
def gpu(mj):
    """libmi355zk bound to cuda:0; fails loudly when the HIP extension or the GPU is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from importlib import import_module
    lib = import_module("mpc-jellyfish_amd.lib")
    lib.init(0)
    return lib


# ---- helpers shared by the tests -------------------------------------------------------------------
def fr_mont_limbs(c, ints):
    """canonical Python ints -> (n,4) uint64 Montgomery limbs (big-int arithmetic, test side)."""
    R = 1 << 256
    out = np.zeros((len(ints), 4), dtype=np.uint64)
    for i, v in enumerate(ints):
        m = v % c.r * R % c.r
        for j in range(4):
            out[i, j] = (m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_ints(a):
    a = np.asarray(a, dtype=np.uint64)
    a = a.reshape(-1, a.shape[-1])
    return [sum(int(a[i, j]) << (64 * j) for j in range(a.shape[1])) for i in range(a.shape[0])]


def fr_from_mont_limbs(c, a):
    rinv = pow(1 << 256, -1, c.r)
    return [v * rinv % c.r for v in limbs_ints(a)]


def affine_limbs(c, pts):
    """list of (x,y) canonical / None -> (n,2,fq_limbs) uint64 Montgomery; None -> (0,0)."""
    L = c.fq_limbs
    R = 1 << (64 * L)
    out = np.zeros((len(pts), 2, L), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is None:
            continue
        for k in range(2):
            m = p[k] * R % c.q
            for j in range(L):
                out[i, k, j] = (m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def affine_from_limbs(c, xy):
    """(2,fq_limbs) Montgomery -> (x,y) canonical or None."""
    v = limbs_ints(np.asarray(xy).reshape(2, c.fq_limbs))
    if v[0] == 0 and v[1] == 0:
        return None
    rinv = pow(1 << (64 * c.fq_limbs), -1, c.q)
    return (v[0] * rinv % c.q, v[1] * rinv % c.q)


def jacobian_to_affine_ints(c, xyz):
    """(3,fq_limbs) Montgomery Jacobian -> canonical affine (x,y) or None, by big-int arithmetic."""
    v = limbs_ints(np.asarray(xyz).reshape(3, c.fq_limbs))
    rinv = pow(1 << (64 * c.fq_limbs), -1, c.q)
    X, Y, Z = (t * rinv % c.q for t in v)
    if Z == 0:
        return None
    zi = pow(Z, -1, c.q)
    return (X * zi * zi % c.q, Y * zi * zi * zi % c.q)


def golden_pt(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def build_circuit(c, log_n, rng, reserved=None):
    """selectors (13 x n), sigma values (5 x n), k, wires (5 x n), public input (n): gates on every 4th
    row family as in test_plonk_gpu, copy constraints as 3-cycles between free cells.
    reserved: {row: value} -- proof-linking gates (relation/src/gates: a(x) * 0 = 0, every selector zero) holding `value`
    on wire 0 of that row."""
    n, r = 1 << log_n, c.r
    reserved = reserved or {}
    k = [1, 7, 13, 17, 23]
    w_n = c.root_of_unity(log_n)
    w = [[rng.randrange(r) for _ in range(n)] for _ in range(5)]
    sel = [[0] * n for _ in range(13)]
    free = []
    for i in range(n):
        kind = 3 if i in reserved else i % 4
        if kind == 0:
            sel[0][i] = sel[1][i] = 1; sel[10][i] = 1
            w[4][i] = (w[0][i] + w[1][i]) % r
            free += [(2, i), (3, i)]
        elif kind == 1:
            sel[4][i] = 3; sel[5][i] = 1; sel[10][i] = 1
            w[4][i] = (3 * w[0][i] * w[1][i] + w[2][i] * w[3][i]) % r
        elif kind == 2:
            sel[6][i] = 1; sel[9][i] = 2; sel[10][i] = 1
            w[4][i] = (pow(w[0][i], 5, r) + 2 * pow(w[3][i], 5, r)) % r
            free += [(1, i), (2, i)]
        elif i in reserved:
            w[0][i] = reserved[i] % r
            free += [(j, i) for j in range(1, 5)]
        else:
            free += [(j, i) for j in range(5)]          # no gate on this row: every cell is free
    pi = [0] * n
    # public input on row 3: q_c + pi + ... = 0 with all selectors 0 except q_lc0 = ... keep it simple: pi = -q_c
    sel[11][3] = 5
    pi[3] = r - 5
    ident = [[k[i] * pow(w_n, j, r) % r for j in range(n)] for i in range(5)]
    perm = {(i, j): (i, j) for i in range(5) for j in range(n)}
    rng.shuffle(free)
    for q in range(0, len(free) - 2, 3):
        a, b, d = free[q], free[q + 1], free[q + 2]
        perm[a], perm[b], perm[d] = b, d, a
        v = rng.randrange(r)
        for (i, j) in (a, b, d):
            w[i][j] = v
    sigma_vals = [[ident[perm[(i, j)][0]][perm[(i, j)][1]] for j in range(n)] for i in range(5)]
    return sel, sigma_vals, k, w, pi




def build_ultra_circuit(c, log_n, rng, range_bits=3):
    """UltraPlonk instance: selectors (14 x n, q_lookup last), sigma values (6 x n), k (6), wires (6 x n), public
    input (n) and the Plookup tables {"range","key","table_dom_sep","q_dom_sep"} (n values each).
    Rows [R, R+T): q_lookup = 1 -- each holds one table entry (domain separator, key, wires 3 and 4) and one lookup
    (wires 0-2 with q_dom_sep) of some entry of that table; rows elsewhere carry the arithmetic gates of build_circuit
    on wires 0-4 and a range-checked value on wire 5 (constraint_system.rs:1441-1480)."""
    n, r = 1 << log_n, c.r
    R = 1 << range_bits
    T = n // 4
    assert R + T < n - 1
    k = [1, 7, 13, 17, 23, 29]
    w_n = c.root_of_unity(log_n)
    w = [[rng.randrange(r) for _ in range(n)] for _ in range(6)]
    sel = [[0] * n for _ in range(14)]
    plookup = {"range": list(range(R)) + [0] * (n - R), "key": [0] * n, "table_dom_sep": [0] * n, "q_dom_sep": [0] * n}
    free = []
    lookup_rows = range(R, R + T)
    for i in lookup_rows:
        sel[13][i] = 1
        plookup["table_dom_sep"][i] = 1 + (i % 2)
        plookup["key"][i] = i - R
    for i in range(n):
        w[5][i] = rng.randrange(R)                          # range wire: every row but the last is looked up in the range table
        if i in lookup_rows:
            tgt = rng.choice(lookup_rows)                   # this row's lookup refers to the table entry at row tgt
            w[5][i] = 0
            plookup["q_dom_sep"][i] = plookup["table_dom_sep"][tgt]
            w[0][i] = plookup["key"][tgt]
            free.append((tgt, i))                           # remembered: values copied below once the table values are final
            continue
        kind = i % 4
        if kind == 0:
            sel[0][i] = sel[1][i] = 1; sel[10][i] = 1
            w[4][i] = (w[0][i] + w[1][i]) % r
        elif kind == 1:
            sel[4][i] = 3; sel[5][i] = 1; sel[10][i] = 1
            w[4][i] = (3 * w[0][i] * w[1][i] + w[2][i] * w[3][i]) % r
        elif kind == 2:
            sel[6][i] = 1; sel[9][i] = 2; sel[10][i] = 1
            w[4][i] = (pow(w[0][i], 5, r) + 2 * pow(w[3][i], 5, r)) % r
    for tgt, i in free:
        w[1][i], w[2][i] = w[3][tgt], w[4][tgt]
    pi = [0] * n
    sel[11][3] = 5
    pi[3] = r - 5
    # copy constraints: 3-cycles among the ungated cells of rows = 3 mod 4 outside the lookup rows, and equal wire-5 values
    ident = [[k[i] * pow(w_n, j, r) % r for j in range(n)] for i in range(6)]
    perm = {(i, j): (i, j) for i in range(6) for j in range(n)}
    cells = [(j, i) for i in range(n) if i % 4 == 3 and i not in lookup_rows for j in range(5)]
    rng.shuffle(cells)
    for q in range(0, len(cells) - 2, 3):
        a, b, d = cells[q], cells[q + 1], cells[q + 2]
        perm[a], perm[b], perm[d] = b, d, a
        v = rng.randrange(r)
        for (i, j) in (a, b, d):
            w[i][j] = v
    by_val = {}
    for j in range(n):
        by_val.setdefault(w[5][j], []).append((5, j))
    for cs in by_val.values():
        for q in range(len(cs)):
            perm[cs[q]] = cs[(q + 1) % len(cs)]
    sigma_vals = [[ident[perm[(i, j)][0]][perm[(i, j)][1]] for j in range(n)] for i in range(6)]
    return sel, sigma_vals, k, w, pi, plookup


def verifying_key(mj, pc, pk, num_inputs):
    """VerifyingKey of preprocess (snark.rs:562-594) as the verifier restatement takes it."""
    pt = lambda cm: None if cm.is_infinity() else affine_from_limbs(pc, cm.xy)
    sel, sig = pk.vk_commitments()
    vk = {"domain_size": pk.n, "num_inputs": num_inputs, "k": list(pk.k), "selector_comms": [pt(x) for x in sel],
          "sigma_comms": [pt(x) for x in sig], "plookup": None}
    if pk.ultra:
        names = ("range_table_comm", "key_table_comm", "table_dom_sep_comm", "q_dom_sep_comm")
        vk["plookup"] = dict(zip(names, [pt(x) for x in pk.plookup_vk_commitments()]))
    return vk
