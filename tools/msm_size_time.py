import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
for cid in (1, 0):
    c = mj.params.CURVES[cid]
    N = 1 << 20
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, N + 2)
    sc = torch.from_numpy(mj.params.random_fr_mont(c, N + 3, seed=3).view(np.int64)).cuda()
    for ln in (14, 16, 17, 18, 19, 20):
        n = (1 << ln) + 3
        s = sc[:n].contiguous()
        mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / 3 * 1e3
        polys = [s] * 5
        mj.kzg.msm_bigint_batch(ck, polys, scalars_are_mont=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mj.kzg.msm_bigint_batch(ck, polys, scalars_are_mont=True)
        torch.cuda.synchronize()
        print("curve", cid, "log", ln, "single ms", round(one, 3), "batch5 ms", round((time.perf_counter() - t0) * 1e3, 3), mj.lib.msm_last_shape())
    ck.release()
