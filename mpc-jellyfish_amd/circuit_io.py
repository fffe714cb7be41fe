"""Circuit files for the compiled host (`mzk_prove <curve> file <path>`, host/mzk_prover.hpp `BenchCircuitHost::read`): a finalised circuit
as the arrays `Arithmetization` exposes (relation/src/constraint_system.rs:1162-1259) -- selector, extended-permutation (and Plookup
table) VALUES on the gate domain H, the wire values witness[wire_variable(i, j)], the coset representatives k, the public input and
the rows it sits on.  Everything little-endian; field elements are 4 x u64 Montgomery limbs (the in-memory image of ark-ff's Fp).

    "MZKCIRC1" | u32 curve_id | u32 num_wire_types (5 | 6) | u32 log_n | u32 n_pub
    k[W] | selectors[nsel][n] | sigmas[W][n] | (W == 6: range, key, table_dom_sep, q_dom_sep [4][n]) | wires[W][n]
    pub_rows[n_pub] (u64) | pub_values[n_pub]
"""
from __future__ import annotations

import struct

import numpy as np

from .params import curve as _curve, fr_to_mont

MAGIC = b"MZKCIRC1"
TABLES = ("range", "key", "table_dom_sep", "q_dom_sep")


def _limbs(c, rows):
    """ints (nested lists) or ready (.., 4) uint64 Montgomery limbs -> contiguous bytes"""
    if isinstance(rows, np.ndarray) and rows.dtype == np.uint64:
        return np.ascontiguousarray(rows).tobytes()
    flat = [int(x) % c.r for row in rows for x in (row if isinstance(row, (list, tuple)) else [row])]
    return fr_to_mont(c, flat).tobytes()


def write_circuit(path: str, curve, log_n: int, selector_values, sigma_values, k, wire_values, pub_input=(), pub_rows=None, tables=None) -> None:
    """selector_values: nsel x n, sigma_values / wire_values: W x n, k: W, tables: {"range", "key", "table_dom_sep", "q_dom_sep"} -> n values
    (UltraPlonk) -- Python ints or uint64 Montgomery limb arrays.  pub_rows None: the public input sits on rows 0 .. len - 1."""
    c = _curve(curve)
    W, n = len(k), 1 << log_n
    assert W in (5, 6) and (W == 6) == (tables is not None) and len(sigma_values) == W and len(wire_values) == W
    assert len(selector_values) == (14 if W == 6 else 13)
    pub_input = list(pub_input)
    rows = list(range(len(pub_input))) if pub_rows is None else list(pub_rows)
    assert len(rows) == len(pub_input) and all(0 <= r < n for r in rows)
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<IIII", c.curve_id, W, log_n, len(pub_input)))
        f.write(_limbs(c, list(k)))
        f.write(_limbs(c, selector_values))
        f.write(_limbs(c, sigma_values))
        if tables is not None:
            f.write(_limbs(c, [tables[t] for t in TABLES]))
        f.write(_limbs(c, wire_values))
        f.write(np.asarray(rows, dtype="<u8").tobytes())
        f.write(_limbs(c, pub_input))


def read_circuit(src) -> dict:
    """The inverse of write_circuit: a path or the file's bytes -> {"curve_id", "num_wire_types", "log_n", "k" (W,4), "selectors" (nsel,n,4),
    "sigmas" (W,n,4), "tables" {name: (n,4)} | None, "wires" (W,n,4), "pub_rows" [..], "pub_values" (n_pub,4)}, uint64 Montgomery limbs."""
    data = src if isinstance(src, (bytes, bytearray)) else open(src, "rb").read()
    if data[:8] != MAGIC:
        raise ValueError("not a MZKCIRC1 circuit file")
    curve_id, W, log_n, n_pub = struct.unpack_from("<IIII", data, 8)
    if W not in (5, 6):
        raise ValueError("num_wire_types must be 5 or 6")
    n, nsel, pos = 1 << log_n, 14 if W == 6 else 13, 24

    def take(rows, cols):
        nonlocal pos
        cnt = rows * cols * 4
        a = np.frombuffer(data, dtype="<u8", count=cnt, offset=pos).astype(np.uint64).reshape(rows, cols, 4)
        pos += cnt * 8
        return a

    out = {"curve_id": curve_id, "num_wire_types": W, "log_n": log_n, "k": take(1, W)[0], "selectors": take(nsel, n), "sigmas": take(W, n)}
    out["tables"] = dict(zip(TABLES, take(4, n))) if W == 6 else None
    out["wires"] = take(W, n)
    out["pub_rows"] = [int(x) for x in np.frombuffer(data, dtype="<u8", count=n_pub, offset=pos)]
    pos += 8 * n_pub
    out["pub_values"] = take(1, n_pub)[0] if n_pub else np.zeros((0, 4), dtype=np.uint64)
    if pos != len(data):
        raise ValueError("trailing or missing bytes in the circuit file")
    return out
