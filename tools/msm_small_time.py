"""MSM latency at proof-sized SRS (SRS length = MSM length + 3, as in prove): wall time per call and device phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
for cid in (0, 1):
    c = mj.params.CURVES[cid]
    for ln in (10, 12, 13, 14, 15, 16, 17):
        n = 1 << ln
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
        s = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=3).view(np.int64)).cuda()
        for _ in range(3):
            mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        torch.cuda.synchronize()
        L.mzk_profile_reset(); L.mzk_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(5):
            mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5 * 1e3
        L.mzk_profile_enable(0)
        ph = {k: round(mlib.profile_get(k)[0] / 5, 3) for k in ("msm_total", "msm_sort", "msm_accumulate", "msm_long", "msm_reduce")}
        t0 = time.perf_counter()
        for _ in range(5):
            mj.kzg.msm_bigint_batch(ck, [s] * 5, scalars_are_mont=True)
        torch.cuda.synchronize()
        wall5 = (time.perf_counter() - t0) / 5 * 1e3
        print("curve", cid, "log", ln, "wall %.3f ms" % wall, "batch5 %.3f ms" % wall5, ph, mlib.msm_last_shape(), flush=True)
        ck.release()
