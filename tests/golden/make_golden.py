"""tests/golden/make_golden.py -- generates the committed fixtures from oracle/pyref.py
(definition-level big-int arithmetic: O(N^2) Horner-evaluation NTT, double-and-add MSM).

The reference holds no golden vectors for this path and cannot be built here,
so these vectors pin the C oracle and the HIP library to the mathematical definitions
(SURVEY.md Appendix B).  Values are canonical integers in hex; tests convert to Montgomery limbs.

    python tests/golden/make_golden.py      # rewrites ntt_vectors.json, msm_vectors.json, kzg_vectors.json
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as P  # noqa: E402

SEED = 0x6d7a6b5f32303236


def hx(v):
    return "%x" % v


def ntt_vectors():
    rng = random.Random(SEED)
    out = []
    for c in (P.BLS12_381, P.BN254):
        for log_n in (0, 1, 2, 3, 4, 5, 6, 8, 10):
            n = 1 << log_n
            for in_len in sorted({n, max(1, n // 2 + 1), max(1, n // 8)}):
                if (log_n == 8 and in_len not in (n, n // 8)) or (log_n == 10 and in_len != n // 8):
                    continue
                coeffs = [rng.randrange(c.r) for _ in range(in_len)]
                if in_len > 2:
                    coeffs[1] = 0
                    coeffs[in_len // 2] = c.r - 1
                for offset in (1, c.fr_gen) if log_n < 10 else (c.fr_gen,):
                    fwd = P.ntt_def(c, coeffs, log_n, offset) if log_n <= 6 else P.ntt_fast(c, coeffs, log_n, offset)
                    inv = P.intt_def(c, coeffs, log_n, offset) if log_n <= 6 else P.ntt_fast(c, coeffs, log_n, offset, inverse=True)
                    out.append({"curve": c.curve_id, "log_n": log_n, "offset": hx(offset),
                                "input": [hx(v) for v in coeffs],
                                "forward": [hx(v) for v in fwd], "inverse": [hx(v) for v in inv]})
    return out


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def msm_vectors():
    rng = random.Random(SEED + 1)
    out = []
    for c in (P.BLS12_381, P.BN254):
        G = P.g1_gen(c)
        for n in (1, 2, 3, 7, 31, 32, 33, 64):
            base_scalars = [rng.randrange(1, c.r) for _ in range(n)]
            bases = [P.g1_mul(c, s, G) for s in base_scalars]
            if n >= 3:
                bases[2] = bases[1]                           # repeated base (P + P in one bucket)
            if n >= 7:
                bases[5] = P.g1_neg(c, bases[4])              # P + (-P) -> infinity
            if n >= 32:
                bases[9] = None                               # point at infinity in the SRS
            edge = [0, 1, 2, c.r - 1, c.r - 2, (1 << 3) - 1, 1 << 3, (1 << 4) - 1, 1 << 4, (1 << 15) - 1, 1 << 15,
                    (1 << 16) - 1, 1 << 16, (1 << 16) + 1, (1 << 32) - 1, 1 << 32, (1 << 128) - 1, 1 << 128,
                    (1 << 240) + 1, (1 << 252) - 1, c.r >> 1, (c.r >> 1) + 1]
            scalars = [rng.randrange(c.r) for _ in range(n)]
            for i, e in enumerate(edge):
                if i < n and (n in (31, 32, 33, 64) or i < 2):
                    scalars[(i * 3) % n] = e
            if n >= 7:
                scalars[4] = scalars[5]                       # k*P + k*(-P) cancels exactly
            res = P.msm_def(c, bases, scalars)
            out.append({"curve": c.curve_id, "bases": [pt(b) for b in bases],
                        "scalars": [hx(s) for s in scalars], "result": pt(res)})
        # all-zero scalars, and a non-reduced 256-bit integer scalar (msm_bigint takes plain integers)
        bases = [P.g1_mul(c, 5, G), P.g1_mul(c, 9, G)]
        out.append({"curve": c.curve_id, "bases": [pt(b) for b in bases], "scalars": ["0", "0"], "result": None})
        big = [(1 << 256) - 1, c.r + 5]
        out.append({"curve": c.curve_id, "bases": [pt(b) for b in bases], "scalars": [hx(s) for s in big],
                    "result": pt(P.msm_def(c, bases, big))})
    return out


def kzg_vectors():
    """Trapdoor KAT: commit(p) over [beta^i]G equals [p(beta)]G."""
    rng = random.Random(SEED + 2)
    out = []
    for c in (P.BLS12_381, P.BN254):
        beta = rng.randrange(c.r)
        srs = P.srs_powers(c, beta, 12)
        for coeffs in ([rng.randrange(c.r) for _ in range(10)],
                       [0, 0, 0] + [rng.randrange(c.r) for _ in range(5)],      # leading zeros skipped (mod.rs:382-386)
                       [rng.randrange(c.r) for _ in range(4)] + [0, 0],          # trailing zeros: degree 3
                       [0] * 6, [7]):
            com = P.g1_mul(c, P.poly_eval(c, coeffs, beta), P.g1_gen(c))
            assert com == P.msm_def(c, srs, coeffs)
            out.append({"curve": c.curve_id, "beta": hx(beta), "srs": [pt(p) for p in srs],
                        "coeffs": [hx(v) for v in coeffs], "commitment": pt(com)})
    return out


if __name__ == "__main__":
    assert P.self_check()
    for name, fn in (("ntt_vectors", ntt_vectors), ("msm_vectors", msm_vectors), ("kzg_vectors", kzg_vectors)):
        data = fn()
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(name, len(data), "cases", os.path.getsize(os.path.join(HERE, name + ".json")) // 1024, "KiB")
