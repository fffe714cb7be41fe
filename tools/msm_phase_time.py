import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
for cid, sizes in ((1, (14, 16, 18, 20)), (0, (14, 16, 18, 20))):
    c = mj.params.CURVES[cid]
    N = 1 << 20
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, N + 2)
    sc = torch.from_numpy(mj.params.random_fr_mont(c, N + 3, seed=3).view(np.int64)).cuda()
    for ln in sizes:
        s = sc[:(1 << ln)].contiguous()
        mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        L.mzk_profile_reset(); L.mzk_profile_enable(1)
        for _ in range(3):
            mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        torch.cuda.synchronize()
        L.mzk_profile_enable(0)
        out = {k: round(mlib.profile_get(k)[0] / 3, 3) for k in ("msm_total", "msm_sort", "msm_accumulate", "msm_long", "msm_reduce")}
        print("curve", cid, "log", ln, out, mlib.msm_last_shape())
    ck.release()
