"""Small MSMs (2^10 .. 2^16 pairs, both curves): the table path (fixed-base table, one shared bucket set of 2^(c-1) buckets, c = 15 .. 17)
against the plain path (no table: c = log2 n - 2, a bucket set per window) -- one call and a batch of five, warm.  Decides PRE_MIN_N
(csrc/msm.hip): below which size an MSM over a registered SRS stays on the plain path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
sizes = [int(a) for a in sys.argv[1:]] or [10, 11, 12, 13, 14, 15, 16]
for cid in (0, 1):
    c = mj.params.CURVES[cid]
    for ln in sizes:
        n = 1 << ln
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
        s = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=3).view(np.int64)).cuda()
        row = []
        for pre in (1, 0):
            L.mzk_msm_set_precompute(pre)
            for _ in range(3):
                mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
                mj.kzg.msm_bigint_batch(ck, [s] * 5, scalars_are_mont=True)
            torch.cuda.synchronize()
            best1 = best5 = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(10):
                    mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
                best1 = min(best1, (time.perf_counter() - t0) / 10 * 1e3)
                shape = mlib.msm_last_shape()
                t0 = time.perf_counter()
                for _ in range(10):
                    mj.kzg.msm_bigint_batch(ck, [s] * 5, scalars_are_mont=True)
                best5 = min(best5, (time.perf_counter() - t0) / 10 * 1e3)
            row.append("%s one %.3f ms batch5 %.3f ms %s" % ("table" if pre else "plain", best1, best5, shape))
        print("curve", cid, "log", ln, " | ".join(row), flush=True)
        L.mzk_msm_set_precompute(1)
        ck.release()
