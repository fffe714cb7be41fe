"""One MSM size under a profiler: python tools/msm_one.py <curve_id> <log_n> [reps]  (SRS length = n + 3 as in prove)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
cid, ln = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
c = mj.params.CURVES[cid]
n = 1 << ln
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
s = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=3).view(np.int64)).cuda()
for _ in range(3):
    mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
torch.cuda.synchronize()
print("curve", cid, "log", ln, "wall %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3))
