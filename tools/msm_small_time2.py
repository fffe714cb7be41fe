"""MSM latency at proof-sized SRS for one window choice (MZK_PRE_C) and sizes given on the command line: single call and batch of 5 (warm)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
sizes = [int(a) for a in sys.argv[1:]] or [15, 16, 17, 18]
for cid in (0, 1):
    c = mj.params.CURVES[cid]
    for ln in sizes:
        n = 1 << ln
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
        s = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=3).view(np.int64)).cuda()
        for _ in range(3):
            mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
            mj.kzg.msm_bigint_batch(ck, [s] * 5, scalars_are_mont=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
        wall = (time.perf_counter() - t0) / 10 * 1e3
        t0 = time.perf_counter()
        for _ in range(10):
            mj.kzg.msm_bigint_batch(ck, [s] * 5, scalars_are_mont=True)
        wall5 = (time.perf_counter() - t0) / 10 * 1e3
        print("MZK_PRE_C", os.environ.get("MZK_PRE_C", "default"), "curve", cid, "log", ln, "one %.3f ms" % wall, "batch5 %.3f ms" % wall5, mlib.msm_last_shape(), flush=True)
        ck.release()
