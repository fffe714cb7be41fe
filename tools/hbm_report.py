#!/usr/bin/env python3
"""tools/hbm_report.py [--log-n 20] [--dense] [--ultra] -- what ONE prover of 2^log_n gates holds in HBM, buffer by buffer: the handle
(coefficient forms, class evaluations, workspace), the keys and their fixed-base tables, and the library's grow-only scratch after a few
proofs (MZK_WS_DEBUG=1 makes mzk_workspace_hbm_bytes print every buffer).  One JSON line on stdout, the per-buffer lines on stderr."""
import argparse
import ctypes as C
import json
import os
import sys

os.environ["MZK_WS_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mpc_jellyfish_amd as mj  # noqa: E402
from importlib import import_module  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--dense", action="store_true")
ap.add_argument("--ultra", action="store_true")
a = ap.parse_args()
mlib = import_module("mpc-jellyfish_amd.lib")
native = mj.snark                                      # (preprocess / prove: thin clients of the round-level C ABI)
L = mlib.init(0)
curve = mj.params.BN254 if a.ultra else mj.params.BLS12_381
n = 1 << a.log_n
rng = mj.rng.test_rng()
ck = mj.UnivariateProverParam.gen_srs_for_testing(curve, mj.rng.fr_rand(curve, rng), n + 2)
cs = mj.snark.gen_circuit_for_bench(curve, n, "UltraPlonk" if a.ultra else "TurboPlonk", **({"dense_seed": 77} if a.dense else {}))
pk = native.preprocess(ck, cs, lagrange=False if a.dense else None)
mlib.check(L.mzk_workspace_release(), "mzk_workspace_release")
import time
for _ in range(3):
    native.prove(rng, cs, pk)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    native.prove(rng, cs, pk)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
hbm = dict(pk.hbm_bytes())
for name, key in (("commit_key", pk.ck), ("lagrange_key", pk.lagrange_ck)):
    if key is not None:
        x, y = C.c_uint64(), C.c_uint64()
        mlib.check(L.mzk_srs_hbm_bytes(key.handle, C.byref(x), C.byref(y)), "mzk_srs_hbm_bytes")
        hbm[name + "_points"], hbm[name + "_fixed_base_table"] = x.value, y.value
x = C.c_uint64()
mlib.check(L.mzk_workspace_hbm_bytes(C.byref(x)), "mzk_workspace_hbm_bytes")
hbm["library_scratch"] = x.value
hbm["total"] = sum(hbm.values())
print(json.dumps({"log_n": a.log_n, "curve": curve.name, "dense_witness": a.dense, "prove_ms": round(ms, 2), "hbm_bytes": hbm,
                  "hbm_total_gb": round(hbm["total"] / 1e9, 2)}))
