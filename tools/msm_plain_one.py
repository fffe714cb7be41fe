"""One VARIABLE-BASE MSM (fixed-base table off: bench.py's headline path) under a profiler:
python tools/msm_plain_one.py <curve_id> <log_n> [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
cid, ln = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
c = mj.params.CURVES[cid]
n = 1 << ln
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n)
mlib.load().mzk_msm_set_precompute(0)
s = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=3).view(np.int64)).cuda()
for _ in range(3):
    mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
torch.cuda.synchronize()
print("curve", cid, "log", ln, "wall %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3))
