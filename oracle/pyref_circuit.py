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
