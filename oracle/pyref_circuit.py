"""oracle/pyref_circuit.py -- TEST INFRASTRUCTURE ONLY.

Loop-by-loop restatement of how the reference builds and finalises its benchmark circuit, to check the vectorised
builder in mpc-jellyfish_amd/snark.py:

    gen_circuit_for_bench            plonk/benches/bench.rs:29-46
    PlonkCircuit::new                relation/src/constraint_system.rs:195-225   (variables 0, 1 and their constant gates)
    Circuit::add -> AdditionGate     relation/src/gadgets/arithmetic.rs, gates/arithmetic.rs:34-51
    finalize_for_arithmetization     constraint_system.rs:966-999; pad :675-685; place_gates proof_linking/linkable_circuit.rs:294-314
    compute_wire_permutation         constraint_system.rs:743-778
    compute_extended_(id_)permutation constraint_system.rs:913-960
PARITY UNPINNED by reference vectors (none exist).
"""
from __future__ import annotations


def bench_circuit(c, num_gates, ultra, range_bit_len, k):
    """Returns (n, wire variable table W x n, witness list, selectors nsel x n, sigma values W x n, tables or None)."""
    r = c.r
    W = 6 if ultra else 5
    witness = [0, 1]
    gates = []                     # (q_lc[4], q_o, q_c)
    wires = [[] for _ in range(W)]

    def insert_gate(wv, q_lc, q_o, q_c):
        gates.append((q_lc, q_o, q_c))
        for i in range(5):
            wires[i].append(wv[i])

    insert_gate([0, 0, 0, 0, 0], [0, 0, 0, 0], 1, 0)          # enforce_constant(0, zero)
    insert_gate([0, 0, 0, 0, 1], [0, 0, 0, 0], 1, 1)          # enforce_constant(1, one)
    a = 0
    for _ in range(num_gates - 10):
        witness.append((witness[a] + witness[1]) % r)
        cvar = len(witness) - 1
        insert_gate([a, 1, 0, 0, cvar], [1, 1, 0, 0], 1, 0)
        a = cvar
    if ultra:
        need = max(len(gates), (1 << range_bit_len) + 1)
    else:
        need = len(gates)
    n = 1
    while n < need:
        n <<= 1
    while len(gates) < n:
        gates.append(([0, 0, 0, 0], 0, 0))                    # PaddingGate
    for i in range(W):
        wires[i] += [0] * (n - len(wires[i]))
    sel = [[0] * n for _ in range(14 if ultra else 13)]
    for j, (q_lc, q_o, q_c) in enumerate(gates):
        for i in range(4):
            sel[i][j] = q_lc[i]
        sel[10][j] = q_o
        sel[11][j] = q_c
    # wire permutation
    occ = [[] for _ in range(len(witness))]
    for i in range(W):
        for j, v in enumerate(wires[i]):
            occ[v].append((i, j))
    perm = {}
    for lst in occ:
        for q, cell in enumerate(lst):
            perm[cell] = lst[(q + 1) % len(lst)]
    w_n = c.root_of_unity(n.bit_length() - 1)
    pw = [1] * n
    for j in range(1, n):
        pw[j] = pw[j - 1] * w_n % r
    sigma = [[k[perm[(i, j)][0]] * pw[perm[(i, j)][1]] % r for j in range(n)] for i in range(W)]
    tables = None
    if ultra:
        tables = {"range": list(range(1 << range_bit_len)) + [0] * (n - (1 << range_bit_len)), "key": [0] * n,
                  "table_dom_sep": [0] * n, "q_dom_sep": [0] * n}
    return n, wires, witness, sel, sigma, tables


# ---- general circuits: every gate family of the hot path, a non-zero public input, copy constraints, lookups ------------------------
# Not a restatement of a reference builder: an ARBITRARY finalised circuit in the arrays `Arithmetization` exposes
# (relation/src/constraint_system.rs:1162-1259), shared by the GPU tests and the golden-proof generator.  Selector order:
# q_lc[0..3], q_mul[0..1], q_hash[0..3], q_o, q_c, q_ecc (+ q_lookup): `all_selectors`, constraint_system.rs:888-905.
def general_circuit(c, log_n, rng, reserved=None):
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




def general_ultra_circuit(c, log_n, rng, range_bits=3):
    """UltraPlonk instance: selectors (14 x n, q_lookup last), sigma values (6 x n), k (6), wires (6 x n), public
    input (n) and the Plookup tables {"range","key","table_dom_sep","q_dom_sep"} (n values each).
    Rows [R, R+T): q_lookup = 1 -- each holds one table entry (domain separator, key, wires 3 and 4) and one lookup
    (wires 0-2 with q_dom_sep) of some entry of that table; rows elsewhere carry the arithmetic gates of general_circuit
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
