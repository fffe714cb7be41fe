"""tools/soak.py -- time-bounded RANDOMISED differential run of the C ABI against the oracle (test infrastructure; run on the GPU box):
    python tools/soak.py [--seconds 300] [--seed 1] [--out gpurun_out/soak.json]

Every case draws its own shape from one seeded generator, so a failure is reproducible from (--seed, case number), which the report
lists.  What a case can be:
  msm     one MSM: curve, 1 .. 2^17 pairs (log-uniform), a base offset into a 2^17 + 3-point SRS, scalars uniform / witness-like small
          / sparse / all equal / top digits at their maximum / window-boundary values (2^(kc) +- 1, 2^(c-1)) / zeros, Montgomery or
          canonical, host or device scalars, the fixed-base table on or off -- against oracle/cpu_ref.c's Pippenger (ark-ec's window rule)
  batch   2 .. 8 MSMs in one mzk_msm_batch{,_dev} call (the fused small-batch path and the grouped large one), ragged lengths and
          offsets, a zero polynomial now and then -- each sum against the oracle
  ntt     curve, log_n 0 .. 19, ragged input length, forward or inverse, plain / GENERATOR coset / random offset -- against the oracle's NTT
  proof   a random GENERAL circuit (oracle/pyref_circuit.py: public input, add / mul / x^5 gates, copy constraints; key and range lookups
          with UltraPlonk) at 2^3 .. 2^9 gates, both curves: the library's rounds (prover.TurboPlonkProver through the round-level ABI)
          against the test-side mirror's bytes, and the restated verifier must accept
  poly    evaluate / divide by X - z / linear combination of device-resident polynomials (rounds 4-5: mzk_poly_{eval,div_linear,lincomb}_dev),
          1 .. 2^17 coefficients, shorter logical lengths, points 0 / 1 / -1 / GENERATOR -- against the oracle's Horner, synthetic division, axpy
The tests under tests/ pin the same paths on fixed seeds; this is the long-running version of them (profiles/r05_soak.json)."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def log_uniform(rng, lo, hi):
    return max(lo, min(hi, int(round(2.0 ** rng.uniform(np.log2(lo), np.log2(hi))))))


def scalar_ints(rng, r, n, shape, c_bits):
    if shape == "uniform":
        return [rng.randrange(r) for _ in range(n)]
    if shape == "small":
        bits = rng.choice([1, 8, 20, 32, 64])
        return [rng.randrange(1 << bits) for _ in range(n)]
    if shape == "sparse":
        return [rng.randrange(r) if rng.random() < 0.05 else 0 for _ in range(n)]
    if shape == "equal":
        return [rng.randrange(r)] * n
    if shape == "top":
        return [r - 1 - rng.randrange(1 << 12) for _ in range(n)]
    if shape == "boundary":
        vals = [0, 1, 2, r - 1, (1 << (c_bits - 1)) - 1, 1 << (c_bits - 1), (1 << c_bits) - 1, 1 << c_bits]
        for k in range(1, 256 // c_bits + 1):
            vals += [((1 << (k * c_bits)) + d) % r for d in (-1, 0, 1)]
        return [rng.choice(vals) for _ in range(n)]
    if shape == "zeros":
        return [0] * n
    raise ValueError(shape)


SHAPES = ["uniform", "uniform", "small", "sparse", "equal", "top", "boundary", "zeros"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak.json"))
    ap.add_argument("--max-log-n", type=int, default=17, help="largest MSM (pairs) and SRS")
    args = ap.parse_args()

    import torch
    import mpc_jellyfish_amd as mj
    from importlib import import_module
    import cref
    import pyref as P
    import pyref_fs as FS
    import pyref_verifier as V
    import mirror_prover as MP
    from conftest import build_circuit, build_ultra_circuit, fr_mont_limbs, verifying_key
    mlib = import_module("mpc-jellyfish_amd.lib")
    L = mlib.init(0)
    cref.lib()

    rng = random.Random(args.seed)
    NMAX = (1 << args.max_log_n) + 3
    srs = {}
    for cid in (0, 1):
        bases = cref.g1_arith_bases(cid, 0x50a0 + cid + args.seed, 0x9e3779b9, NMAX)
        srs[cid] = (bases, mj.UnivariateProverParam.from_affine(cid, bases))
    threads = min(16, len(os.sched_getaffinity(0)))

    def ints_to_limbs(ints):
        a = np.zeros((len(ints), 4), dtype=np.uint64)
        for i, v in enumerate(ints):
            for k in range(4):
                a[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        return a

    def case_msm(case):
        cid = rng.randrange(2)
        c = mj.params.CURVES[cid]
        n = log_uniform(rng, 1, 1 << args.max_log_n)
        off = rng.randrange(0, NMAX - n + 1)
        shape = rng.choice(SHAPES)
        mont, dev, table = rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.7
        ints = scalar_ints(rng, c.r, n, shape, rng.choice([13, 15, 16, 17, 20]))
        sc = fr_mont_limbs(c, ints) if mont else ints_to_limbs(ints)
        desc = {"kind": "msm", "curve": cid, "n": n, "off": off, "shape": shape, "mont": mont, "dev": dev, "table": table}
        bases, pp = srs[cid]
        L.mzk_msm_set_precompute(1 if table else 0)
        arg = torch.from_numpy(sc.view(np.int64)).cuda() if dev else sc
        got = cref.jac_to_affine(cid, mj.msm_bigint(pp, arg, base_offset=off, scalars_are_mont=mont))[0]
        L.mzk_msm_set_precompute(1)
        want = cref.jac_to_affine(cid, cref.msm(cid, bases[off:off + n], sc, scalars_are_mont=mont, threads=threads))[0]
        return desc, bool(np.array_equal(got, want))

    def case_batch(case):
        cid = rng.randrange(2)
        c = mj.params.CURVES[cid]
        count = rng.randrange(2, 9)
        top = log_uniform(rng, 16, 1 << min(args.max_log_n, 16))
        lens, offs, sets = [], [], []
        for k in range(count):
            n = top if rng.random() < 0.5 else rng.randrange(0, top + 1)
            lens.append(n)
            offs.append(rng.randrange(0, min(8, NMAX - n) + 1))
            ints = scalar_ints(rng, c.r, n, rng.choice(SHAPES), 16)
            sets.append(fr_mont_limbs(c, ints) if n else np.zeros((0, 4), dtype=np.uint64))
        dev = rng.random() < 0.6
        desc = {"kind": "batch", "curve": cid, "lens": lens, "offs": offs, "dev": dev}
        bases, pp = srs[cid]
        arg = [torch.from_numpy(s.view(np.int64)).cuda() for s in sets] if dev else sets
        jac = mj.msm_bigint_batch(pp, arg, offs, scalars_are_mont=True)
        ok = True
        for k in range(count):
            want = cref.jac_to_affine(cid, cref.msm(cid, bases[offs[k]:offs[k] + lens[k]], sets[k], scalars_are_mont=True, threads=threads))[0]
            ok = ok and bool(np.array_equal(cref.jac_to_affine(cid, jac[k])[0], want))
        return desc, ok

    def case_ntt(case):
        cid = rng.randrange(2)
        c = mj.params.CURVES[cid]
        log_n = rng.randrange(0, 20)
        n = 1 << log_n
        in_len = n if rng.random() < 0.4 else rng.randrange(0, n + 1)
        inverse = rng.random() < 0.5
        which = rng.choice(["plain", "generator", "random"])
        offset = 1 if which == "plain" else (c.fr_generator if which == "generator" else rng.randrange(2, c.r))
        off_limbs = None if which == "plain" else mj.params.fr_to_mont(c, [offset])[0]
        desc = {"kind": "ntt", "curve": cid, "log_n": log_n, "in_len": in_len, "inverse": inverse, "offset": which}
        x = mj.params.random_fr_mont(c, max(in_len, 1), seed=rng.randrange(1 << 30))[:in_len]
        padded = np.zeros((n, 4), dtype=np.uint64)
        padded[:in_len] = x
        d = mj.Radix2EvaluationDomain(cid, log_n)
        if which != "plain":
            d = d.get_coset(offset)
        got = d.ifft(x) if inverse else d.fft(x)
        want = cref.ntt(cid, padded, log_n, inverse, off_limbs, threads=threads)
        return desc, bool(np.array_equal(got, want))

    TABLES = ("range", "key", "table_dom_sep", "q_dom_sep")

    def case_proof(case):
        cid = rng.randrange(2)
        ultra = rng.random() < 0.5
        log_n = rng.randrange(4 if ultra else 3, 10)
        c, pc = mj.params.CURVES[cid], P.CURVES[cid]
        n, W = 1 << log_n, 6 if ultra else 5
        crng = random.Random(rng.randrange(1 << 60))
        desc = {"kind": "proof", "curve": cid, "ultra": ultra, "log_n": log_n}
        dom = mj.Radix2EvaluationDomain(c, log_n)
        kw = {}
        if ultra:
            sel, sig, k, w, pi, tabs = build_ultra_circuit(pc, log_n, crng)
            kw = {"plookup": {name: dom.ifft(fr_mont_limbs(c, tabs[key])) for name, key in zip(mj.plonk.PLOOKUP_TABLE_POLYS, TABLES)}}
        else:
            sel, sig, k, w, pi = build_circuit(pc, log_n, crng)
        sel_p, sig_p = [dom.ifft(fr_mont_limbs(c, s)) for s in sel], [dom.ifft(fr_mont_limbs(c, s)) for s in sig]
        wires = np.stack([fr_mont_limbs(c, col) for col in w])
        srs_beta = crng.randrange(1, c.r)
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)
        mirror = MP.TurboPlonkProver(c, n, sel_p, sig_p, k, ck, **kw)
        native = mj.prover.TurboPlonkProver(c, n, sel_p, sig_p, k, ck, **kw)
        pub = pi[:4]
        blind = mj.snark.draw_blinders(c, mj.rng.test_rng(), W, ultra)
        want = mj.snark.serialize_proof(c, mirror.prove(wires, fr_mont_limbs(c, pi), mj.prover.TranscriptChallenges(mirror, pub), blind))
        got = mj.snark.serialize_proof(c, native.prove(wires, pub, mj.prover.TranscriptChallenges(native, pub), blind))
        ok = got == want
        if ok and rng.random() < 0.5:                                   # (the pairing check is pure Python: seconds)
            vk = verifying_key(mj, pc, native, len(pub))
            ok = bool(V.verify(pc, FS.StandardTranscript(pc, b"PlonkProof"), vk, pub, got, P.g1_gen(pc), srs_beta))
            desc["verified"] = True
        mirror.release()
        native.release()
        ck.release()
        return desc, ok

    def case_poly(case):
        cid = rng.randrange(2)
        c = mj.params.CURVES[cid]
        n = log_uniform(rng, 1, 1 << 17)
        op = rng.choice(["eval", "div", "lincomb"])
        desc = {"kind": "poly", "curve": cid, "n": n, "op": op}
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
        special = [0, 1, c.r - 1, c.fr_generator]
        x = rng.choice(special) if rng.random() < 0.3 else rng.randrange(c.r)
        xm = mj.params.fr_to_mont(c, [x])[0]
        if op == "eval":
            batch = rng.randrange(1, 4)
            length = n if rng.random() < 0.5 else rng.randrange(0, n + 1)
            desc.update(batch=batch, length=length, x=hex(x))
            polys = mj.params.random_fr_mont(c, batch * n, seed=rng.randrange(1 << 30)).reshape(batch, n, 4)
            got = mj.poly.evaluate(c, dev(polys), x, length=length)
            want = [mj.params.fr_from_mont(c, cref.poly_eval(cid, polys[b][:length], xm).reshape(1, 4))[0] if length else 0 for b in range(batch)]
            return desc, got == want
        if op == "div":
            desc.update(z=hex(x))
            pl = mj.params.random_fr_mont(c, n, seed=rng.randrange(1 << 30))
            got = mj.poly.div_by_linear(c, dev(pl), x).cpu().numpy().view(np.uint64)
            return desc, bool(np.array_equal(got, cref.poly_div_linear(cid, pl, xm)))
        k = rng.randrange(1, 9)
        lens = [rng.randrange(1, n + 1) for _ in range(k)]
        out_len = rng.choice([max(lens), rng.randrange(1, max(lens) + 5)])
        sc = [rng.choice([0, 1, c.r - 1]) if rng.random() < 0.2 else rng.randrange(c.r) for _ in range(k)]
        desc.update(lens=lens, out_len=out_len)
        polys = [mj.params.random_fr_mont(c, ln, seed=rng.randrange(1 << 30)) for ln in lens]
        got = mj.poly.lincomb(c, list(zip(sc, [dev(pl) for pl in polys])), out_len=out_len).cpu().numpy().view(np.uint64)
        return desc, bool(np.array_equal(got, cref.poly_lincomb(cid, polys, mj.params.fr_to_mont(c, sc), out_len)))

    kinds = [("msm", case_msm, 5), ("batch", case_batch, 2), ("ntt", case_ntt, 4), ("proof", case_proof, 1), ("poly", case_poly, 3)]
    weights = [k[2] for k in kinds]
    counts = {k[0]: 0 for k in kinds}
    seconds = {k[0]: 0.0 for k in kinds}
    failures = []
    t_start = time.time()
    last_print = t_start
    case = 0
    while time.time() - t_start < args.seconds:
        name, fn, _ = rng.choices(kinds, weights=weights)[0]
        t0 = time.time()
        try:
            desc, ok = fn(case)
            err = None
        except Exception as e:                                          # an error code from the library on a valid input is a failure too
            desc, ok, err = {"kind": name}, False, repr(e)
        seconds[name] += time.time() - t0
        counts[name] += 1
        if not ok:
            failures.append({"case": case, "desc": desc, "error": err})
            print("MISMATCH", case, desc, err, flush=True)
        case += 1
        if time.time() - last_print > 30:
            last_print = time.time()
            print(f"[soak] {case} cases, {len(failures)} failures, {time.time() - t_start:.0f} s", flush=True)
    for cid in (0, 1):
        srs[cid][1].release()
    report = {"seed": args.seed, "seconds": round(time.time() - t_start, 1), "cases": case, "by_kind": counts,
              "seconds_by_kind": {k: round(v, 1) for k, v in seconds.items()}, "failures": failures, "max_log_n": args.max_log_n,
              "oracle": "oracle/cpu_ref.c (Pippenger with ark-ec's window rule, radix-2 NTT), tests/mirror_prover.py, oracle/pyref_verifier.py -- "
                        "this repo's restatements; parity with the Rust code is unpinned (DESIGN.md section 2)"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: report[k] for k in ("seed", "seconds", "cases", "by_kind")}), "failures:", len(failures))
    return 1 if failures else 0


if __name__ == "__main__":
    raise SystemExit(main())
