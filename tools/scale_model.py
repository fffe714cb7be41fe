#!/usr/bin/env python3
"""tools/scale_model.py -- what `PlonkKzgSnark::prove` should take on G = 1, 2, 4, 8 GPUs, from components MEASURED ON ONE GPU.

The driver measures the real 1 -> 8 curve on a node this builder never sees; this model is what that curve is to be checked
against ("Next round" 6a).  Per configuration (C4: TurboPlonk / BLS12-381 / 2^20 gates; C5: UltraPlonk / BN254 / 2^22)
and per G it measures, on this one GPU, exactly what rank 0 of a G-rank run executes:

  commits      every batch_commit of the proof (round 1: W wires; 1.5: h_1, h_2; 2: z; 2.5: Plookup product; 3: W split-quotient
               parts; 5: two openings) as ONE mzk_msm_batch_dev over the rank's point range [0, len / G) of every polynomial
               (sharding.ShardedCommitter), under THREE schedules (round 4):
                 replicated   every rank holds the whole SRS and its table (window for 2^20+ points) and commits over its range of it
                              (round 3's set-up);
                 sliced       every rank holds only its range (mzk_srs_slice: 1 / G of the table, window chosen for the slice; small
                              shards are fused into one sort / accumulation) -- what the compiled host does since round 4;
                 by_polynomial  the reference's own parallelism (univariate_kzg/mod.rs:119-131, one polynomial per worker): groups of W
                              polynomials as whole polynomials on W ranks, the spare ranks taking point-range halves of the largest;
                              groups of one or two polynomials by point range as above.  Its cost on the busiest rank is measured
                              as the batch that rank would run;
  quotient     the rank's ceil(needed / G) residue classes (needed = W of the 8) through mzk_plonk_quotient_chunked_dev, the top coefficients and
               the inverse-Vandermonde combine every rank runs after the exchange;
  ranged       rounds 4 and 5 (evaluations; linearisation + batch polynomials, their division by (X - z)) on the rank's
               coefficient range only (prover.py _RangeEvals / _openings_ranged): timed in a real proof whose committer reports
               world = G and rank 0's range (RankZero below; the exchanged values are then not the true ones, which no later
               prover stage checks);
  replicated   everything else of a proof, which every rank repeats: wire / z iNTTs and masking, grand products, the Plookup
               sorted vector, quotient split, transcript -- taken from the profiled rounds of a real single-GPU proof;
and it ADDS, from stated constants (not measurable on one GPU):
  collectives  one small all-gather per commit group (k x 144 / 96 bytes per rank) at SMALL_COLLECTIVE_US each, and the one
               exchange of class remainders: (G - 1) x classes_per_rank x n x 32 bytes received per rank at XGMI_GBPS.

    python tools/scale_model.py [--c5-log-n 22] > profiles/r04_scale_model.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# One small collective of a commit group = the wire + the host side of sharding.gather_partials.  The HOST side is measured
# (tools/collective_time.py, profiles/r05_collective_time.json: tensor set-up, one transfer back, k x (G - 1) Jacobian additions through
# mzk_g1_sum_jacobian: 36 us for k = 5 at two ranks, 51 us at four).  The WIRE cannot be measured without a multi-GPU node: the same call
# between gloo processes over TCP loopback takes 0.7 ms at two ranks -- a figure of the test backend, not of RCCL -- so the RCCL
# all-gather of a few hundred bytes over xGMI, device -> host hop included, stays the ASSUMED 40 us.
RCCL_SMALL_ALLGATHER_US = 40.0
HOST_SUMS_US = 45.0
SMALL_COLLECTIVE_US = RCCL_SMALL_ALLGATHER_US + HOST_SUMS_US
XGMI_GBPS = 300.0                 # all-gather receive rate per GPU: 7 links x ~45 GB/s achieved of 64 GB/s per direction (MI355X_MICROARCH.md)


def median_ms(fn, reps=5, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


class RankZero:
    """Stands in for sharding.ShardedCommitter on ONE GPU: commits are whole (so the proof's quotient is the true one and passes the
    degree check), but it reports `world` ranks and rank 0's point range, so rounds 4 and 5 do what rank 0 of a G-rank run does."""

    def __init__(self, mj, ck, world):
        self.mj, self.ck, self.G, self.group = mj, ck, world, None

    def world(self):
        return self.G

    def rank(self):
        return 0

    def point_range(self):
        return 0, self.ck.length // self.G

    def commit_jacobian(self, polys):
        return self.mj.msm_bigint_batch(self.ck, [p.contiguous() for p in polys], scalars_are_mont=True)

    def all_gather_fr(self, values):
        return [list(values)] * self.G

    def commit_jacobian_slices(self, slices):
        return self.mj.msm_bigint_batch(self.ck, [s.contiguous() for s in slices], scalars_are_mont=True)


def model(mj, curve, plonk_type, log_n):
    import torch
    import mirror_prover as MP      # the per-stage hooks this model times (rank 0's share of rounds 4-5 through a stand-in committer, chunked keys)
                                    # belong to the test-side sequencing of the primitives (tests/mirror_prover.py); the product's rounds are in the library
    c = curve
    n = 1 << log_n
    ultra = plonk_type == "UltraPlonk"
    W = 6 if ultra else 5
    cs = mj.snark.gen_circuit_for_bench(c, n, plonk_type)
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), n + 2)
    # ---- one real proof on one GPU: total and per-round times ----
    pk = MP.preprocess(ck, cs)
    for _ in range(3):
        MP.prove(rng, cs, pk)
    torch.cuda.synchronize()
    import gc
    gc.collect()                                                     # (the interpreter's first full collection would land in a timed proof: bench.py)
    gc.freeze()
    t0 = time.perf_counter()
    for _ in range(5):
        MP.prove(rng, cs, pk)
    torch.cuda.synchronize()
    prove1_ms = (time.perf_counter() - t0) / 5 * 1e3
    core, _ = MP.prove(rng, cs, pk, profile=True)
    rounds = dict(core.timings_ms)
    needed = list(pk.classes_needed)
    ranged_keys = ["r4_evals", "r5_polys"]
    ranged_ms = {1: sum(rounds[k] for k in ranged_keys)}
    pk.identity_check = False                                        # (RankZero's exchanged values are stand-ins)
    pk._allow_unchecked = True
    for G in (2, 4, 8):
        pk.committer = RankZero(mj, ck, G)
        samples = []
        for _ in range(5):
            core, _ = MP.prove(rng, cs, pk, profile=True)
            samples.append(sum(dict(core.timings_ms)[k] for k in ranged_keys))
        ranged_ms[G] = sorted(samples)[2]
    pk.committer = None
    pk.release()
    del pk
    torch.cuda.empty_cache()
    replicated_keys = [k for k in rounds if not k.endswith("_commit") and k != "r3_quotient" and k not in ("r4_evals", "r5_polys")]
    replicated_ms = sum(rounds[k] for k in replicated_keys)
    # ---- commit groups: (number of polynomials, length) in proof order ----
    groups = [("r1_wires", W, n + 2)]
    if ultra:
        groups.append(("r1_5_h", 2, n + 3))
    groups.append(("r2_z", 1, n + 3))
    if ultra:
        groups.append(("r2_5_prod_lookup", 1, n + 3))
    groups += [("r3_split_quotient", W, n + 3), ("r5_openings", 2, n + 2)]
    scal = torch.from_numpy(mj.params.random_fr_mont(c, n + 3, seed=5).view(np.int64)).cuda()
    # round 1 is committed from the wire VALUES over the Lagrange-basis key (both hosts from 2^13 gates on; the compiled host shards it by
    # point range like every other commitment): rows of n values + 2 blinders
    lag = ck.lagrange_key(n)
    ext = torch.zeros((W, n + 3, 4), dtype=torch.int64, device="cuda")
    ext[:, :n] = cs.wire_values
    ext[:, n:n + 2] = scal[:2]
    out = {"config": {"plonk_type": plonk_type, "curve": c.name, "log_n": log_n, "classes_needed": len(needed)},
           "measured_one_gpu": {"prove_ms": round(prove1_ms, 2), "rounds_ms": rounds, "replicated_ms": round(replicated_ms, 2),
                                "replicated_stages": replicated_keys},
           "per_G": {}}
    # ---- quotient: chunked keys holding 1, 2, .. classes ----
    class_ms = {}
    slab = None
    for per in sorted({-(-len(needed) // G) for G in (1, 2, 4, 8)}):
        key = MP.preprocess(ck, cs, quotient_classes=needed[:per])
        if slab is None:
            rows = W + 2 + (3 if ultra else 0)
            slab = torch.from_numpy(mj.params.random_fr_mont(c, rows * (n + 3), seed=6).view(np.int64).reshape(rows, n + 3, 4)).cuda()
        ch = mj.plonk.Challenges(0x1234567, 0x89abcde, 0xf012345, 0x1357911)
        res = torch.empty((per, n, 4), dtype=torch.int64, device="cuda")
        class_ms[per] = median_ms(lambda: mj.plonk.compute_quotient_chunked_dev(key.pk, ch, slab, n + 3, out_dev=res, pi_zero=True))
        if per == len(needed):
            top_ms = median_ms(lambda: mj.plonk.compute_quotient_top_dev(key.pk, ch, slab, n + 3))
        key.release()
        del key, res
        torch.cuda.empty_cache()
    rem = torch.from_numpy(mj.params.random_fr_mont(c, len(needed) * n, seed=7).view(np.int64).reshape(len(needed), n, 4)).cuda()
    quot = torch.empty((8 * n, 4), dtype=torch.int64, device="cuda")
    top = torch.from_numpy(mj.params.random_fr_mont(c, 16, seed=8).view(np.int64)).cuda()
    combine_ms = top_ms + median_ms(lambda: mj.plonk.combine_quotient_classes(c, n, rem, classes=needed, out_dev=quot, top=top, n_top=W + 3))
    del rem, quot
    point_bytes = 3 * c.fq_limbs * 8
    for G in (1, 2, 4, 8):
        commits, sliced, by_poly = {}, {}, {}
        ck_s = ck.slice(0, (n + 3) // G) if G > 1 else ck                # rank 0's range as an SRS of its own
        lag_s = lag.slice(0, (n + 3) // G) if G > 1 else lag
        for name, k, length in groups:
            hi = length // G if G > 1 else length                       # rank 0's point range [0, len / G)
            if name == "r1_wires":
                sets = [ext[i, :hi] for i in range(W)]
                commits[name] = round(median_ms(lambda: mj.msm_bigint_batch(lag, sets, scalars_are_mont=True)), 3)
                sliced[name] = round(median_ms(lambda: mj.msm_bigint_batch(lag_s, sets, scalars_are_mont=True)), 3)
                # whole polynomials: rank 0 commits ceil(W / G) wires whole (from their values over the Lagrange key), G >= W: one
                whole = [ext[i, :length] for i in range(-(-W // G))]
                by_poly[name] = round(median_ms(lambda: mj.msm_bigint_batch(lag, whole, scalars_are_mont=True)), 3) if G > 1 else commits[name]
                continue
            sets = [scal[:hi]] * k
            commits[name] = round(median_ms(lambda: mj.msm_bigint_batch(ck, sets, scalars_are_mont=True)), 3)
            sliced[name] = round(median_ms(lambda: mj.msm_bigint_batch(ck_s, sets, scalars_are_mont=True)), 3)
            if k >= W and G > 1:
                # the busiest rank: floor(W / G) whole polynomials and, if G does not divide W, its share of the rest cut by point range
                full, rest = W // G, W % G
                part = [scal[:length]] * full + ([scal[:max(1, length * rest // G)]] if rest and G < W else [])
                if not part:
                    part = [scal[:length]]                               # G > W: W ranks take one whole polynomial each
                by_poly[name] = round(median_ms(lambda: mj.msm_bigint_batch(ck, part, scalars_are_mont=True)), 3)
            else:
                by_poly[name] = sliced[name]                             # one or two polynomials: by point range (on the sliced key)
        if G > 1:
            ck_s.release()
            lag_s.release()
        per = -(-len(needed) // G)
        gather_bytes = (G - 1) * per * n * 32
        exchange_ms = 0.0 if G == 1 else SMALL_COLLECTIVE_US / 1e3 + gather_bytes / (XGMI_GBPS * 1e9) * 1e3
        small_ms = 0.0 if G == 1 else (len(groups) + 2) * SMALL_COLLECTIVE_US / 1e3      # + the round-4 and round-5 exchanges of partial values
        rest_ms = class_ms[per] + combine_ms + exchange_ms + small_ms + replicated_ms + ranged_ms[G]
        best = min((sum(sliced.values()), "sliced"), (sum(by_poly.values()), "by_polynomial"), (sum(commits.values()), "replicated"))
        total = best[0] + rest_ms
        out["per_G"][str(G)] = {"commits_ms": commits, "commit_total_ms": round(sum(commits.values()), 2),
                                "commits_sliced_srs_ms": sliced, "commit_total_sliced_srs_ms": round(sum(sliced.values()), 2),
                                "commits_by_polynomial_ms": by_poly, "commit_total_by_polynomial_ms": round(sum(by_poly.values()), 2),
                                "best_commit_schedule": best[1], "predicted_prove_replicated_srs_ms": round(sum(commits.values()) + rest_ms, 2),
                                "quotient_classes_per_rank": per, "quotient_local_ms": round(class_ms[per], 3), "combine_ms": round(combine_ms, 3),
                                "class_exchange_ms": round(exchange_ms, 3), "class_exchange_bytes_received": gather_bytes,
                                "small_collectives_ms": round(small_ms, 3), "replicated_ms": round(replicated_ms, 2),
                                "ranged_rounds_4_5_ms": round(ranged_ms[G], 3), "predicted_prove_ms": round(total, 2)}
    base = out["per_G"]["1"]["predicted_prove_ms"]
    for G in ("1", "2", "4", "8"):
        out["per_G"][G]["speedup_vs_1"] = round(base / out["per_G"][G]["predicted_prove_ms"], 2)
    out["model_vs_measured_at_G1"] = round(base / prove1_ms, 3)
    lag.release()
    ck.release()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c4-log-n", type=int, default=20)
    ap.add_argument("--c5-log-n", type=int, default=22)
    args = ap.parse_args()
    import mpc_jellyfish_amd as mj
    from importlib import import_module
    import_module("mpc-jellyfish_amd.lib").init(0)
    res = {"what": "predicted PlonkKzgSnark::prove time on G GPUs from single-GPU measurements of each rank's share (tools/scale_model.py)",
           "constants": {"small_collective_us": SMALL_COLLECTIVE_US, "of_which_assumed_rccl_allgather_us": RCCL_SMALL_ALLGATHER_US, "of_which_measured_host_sums_us": HOST_SUMS_US, "xgmi_allgather_GBps_per_gpu": XGMI_GBPS},
           "C4_turbo_bls12_381": model(mj, mj.params.BLS12_381, "TurboPlonk", args.c4_log_n)}
    if args.c5_log_n:
        res["C5_ultra_bn254"] = model(mj, mj.params.BN254, "UltraPlonk", args.c5_log_n)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
