import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
for cid in (0, 1):
    c = mj.params.CURVES[cid]
    for ln in (10, 12, 13, 15, 16, 17, 18, 19, 20):
        N = 1 << ln
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, N + 2)
        s = torch.from_numpy(mj.params.random_fr_mont(c, N + 3, seed=3).view(np.int64)).cuda()
        mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        L.mzk_profile_reset(); L.mzk_profile_enable(1)
        for _ in range(3):
            mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        torch.cuda.synchronize()
        L.mzk_profile_enable(0)
        out = {k[4:]: round(mlib.profile_get(k)[0] / 3, 3) for k in ("msm_total", "msm_sort", "msm_accumulate", "msm_long", "msm_reduce")}
        print("curve", cid, "srs=n=2^%d+3" % ln, out, mlib.msm_last_shape(), flush=True)
        ck.release()
