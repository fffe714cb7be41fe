#!/usr/bin/env python3
"""tools/prove_in_flight.py [--log-n 20] [--in-flight 1,2,3] [--reps 10] [--dense] -- K proofs IN FLIGHT on ONE card.

K host threads, each bound to its own device context of libmi355zk (MZK_VIRTUAL_DEVICES=K maps K contexts onto the one GPU: own
lock, workspace, streams, keys), each proving the same 2^log_n-gate bench circuit through the round-level C ABI (mzk_prover_*) `reps`
times: proofs/s of the card against 1000 / prove_ms of one prover alone.  What overlaps: the host Horner tails, transcript hashing,
blinder draws and the launch gaps of one proof with the kernels of the other -- and, when the provers run on their own streams, the
latency-bound narrow levels of one MSM reduction with the accumulation of the other.
Prints one JSON line.  (The process sets MZK_VIRTUAL_DEVICES itself, before the library is loaded.)"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(log_n, ks, reps, dense=False, plonk_type="TurboPlonk", curve_id=0, check=False):
    os.environ["MZK_VIRTUAL_DEVICES"] = str(max(ks))
    if os.environ.get("MZK_INFLIGHT_HW_QUEUES"):                          # (experiment: more hardware queues for the K x 4 streams; read by HIP at start-up)
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ["MZK_INFLIGHT_HW_QUEUES"]
    import torch
    import mpc_jellyfish_amd as mj
    from importlib import import_module
    mlib = import_module("mpc-jellyfish_amd.lib")
    native = mj.snark                                      # (preprocess / prove: thin clients of the round-level C ABI)
    L = mlib.load()
    curve = mj.params.CURVES[curve_id]
    n = 1 << log_n
    K = max(ks)
    state = [None] * K
    errors = []

    def setup(k):
        try:
            mlib.check(L.mzk_init(k), "mzk_init")
            rng = mj.rng.test_rng()
            beta = mj.rng.fr_rand(curve, rng)
            ck = mj.UnivariateProverParam.gen_srs_for_testing(curve, beta, n + 2)
            cs = mj.snark.gen_circuit_for_bench(curve, n, plonk_type, **({"dense_seed": 77} if dense else {}))
            pk = native.preprocess(ck, cs, lagrange=False if dense else None)
            for _ in range(3):
                native.prove(rng, cs, pk)
            torch.cuda.synchronize()
            state[k] = (rng, cs, pk, ck)
        except Exception as e:                                            # noqa: BLE001
            errors.append("setup %d: %r" % (k, e))

    for k in range(K):                                                    # set-up one after the other (each holds the GPU for ~1 s)
        t = threading.Thread(target=setup, args=(k,))
        t.start()
        t.join()
    if errors:
        raise SystemExit("; ".join(errors))
    out = {}
    digests = set()
    for k_now in ks:
        start = threading.Barrier(k_now + 1)
        done = [0.0] * k_now

        def work(k):
            try:
                mlib.check(L.mzk_init(k), "mzk_init")
                rng, cs, pk, _ = state[k]
                start.wait()
                for _ in range(reps):
                    native.prove(rng, cs, pk)
                torch.cuda.synchronize()
                done[k] = time.perf_counter()
            except Exception as e:                                        # noqa: BLE001
                errors.append("work %d: %r" % (k, e))
                try:
                    start.abort()
                except Exception:                                         # noqa: BLE001
                    pass

        th = [threading.Thread(target=work, args=(k,)) for k in range(k_now)]
        for t in th:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        if errors:
            raise SystemExit("; ".join(errors))
        el = max(done) - t0
        out[k_now] = {"proofs_per_s": round(k_now * reps / el, 2), "ms_per_proof_per_prover": round(el / reps * 1e3, 2)}
    # --check: K threads prove CONCURRENTLY from identical rng streams: every proof made while others are in flight must be the same bytes
    concurrent_ok = None
    if check:
        K2 = max(ks)
        start = threading.Barrier(K2)
        got = [[] for _ in range(K2)]

        def work_check(k):
            try:
                mlib.check(L.mzk_init(k), "mzk_init")
                _, cs, pk, _ = state[k]
                start.wait()
                for _ in range(4):
                    g = mj.rng.test_rng()
                    mj.rng.fr_rand(curve, g)                              # (the SRS trapdoor was the first draw of the stream)
                    got[k].append(hashlib.sha256(native.prove(g, cs, pk)[1]).hexdigest())
            except Exception as e:                                        # noqa: BLE001
                errors.append("check %d: %r" % (k, e))
        th = [threading.Thread(target=work_check, args=(k,)) for k in range(K2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise SystemExit("; ".join(errors))
        concurrent_ok = len({d for row in got for d in row}) == 1 and all(len(row) == 4 for row in got)
    # every context must emit the same bytes for the same rng stream
    for k in range(K):
        def one(k=k):
            mlib.check(L.mzk_init(k), "mzk_init")
            _, cs, pk, _ = state[k]
            digests.add(hashlib.sha256(native.prove(mj.rng.test_rng(), cs, pk)[1]).hexdigest()[:16])
        t = threading.Thread(target=one)
        t.start()
        t.join()
    res = {"log_n": log_n, "plonk_type": plonk_type, "curve": curve.name, "dense_witness": dense, "reps_per_prover": reps,
           "in_flight": {str(k): v for k, v in out.items()}, "contexts_agree_on_proof": len(digests) == 1,
           "proofs_made_concurrently_identical": concurrent_ok}
    if 1 in out:
        for k, v in out.items():
            if k != 1:
                res["gain_%d_in_flight" % k] = round(v["proofs_per_s"] / out[1]["proofs_per_s"], 3)
    for k in range(K):
        def rel(k=k):
            mlib.check(L.mzk_init(k), "mzk_init")
            _, _, pk, ck = state[k]
            if pk.lagrange_ck is not None:
                pk.lagrange_ck.release()
            pk.release()
            ck.release()
        t = threading.Thread(target=rel)
        t.start()
        t.join()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--in-flight", default="1,2,3")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--ultra", action="store_true", help="UltraPlonk over BN254 instead of TurboPlonk over BLS12-381")
    ap.add_argument("--check", action="store_true", help="also: K threads prove concurrently from identical rng streams; all proofs must be the same bytes")
    a = ap.parse_args()
    print(json.dumps(measure(a.log_n, [int(x) for x in a.in_flight.split(",")], a.reps, a.dense, "UltraPlonk" if a.ultra else "TurboPlonk",
                             1 if a.ultra else 0, a.check)))
