"""tools/msm_window_sweep.py -- which window the fixed-base table of a SMALL SRS should use: one MSM of n = |SRS| pairs and a batch
of 5, for every eligible window, per curve and size (run as separate processes: the window is fixed when the table is built).
    for c in 15 16 17 20; do MZK_PRE_C=$c python tools/msm_window_sweep.py; done"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mpc_jellyfish_amd as mj
from importlib import import_module

mlib = import_module("mpc-jellyfish_amd.lib")
L = mlib.ensure_init()
for cid in (0, 1):
    c = mj.params.CURVES[cid]
    for ln in (14, 15, 16, 17, 18, 19):
        N = 1 << ln
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, N + 2)
        s = torch.from_numpy(mj.params.random_fr_mont(c, N + 3, seed=3).view(np.int64)).cuda()
        res = []
        for k in (1, 5):
            sets = [s] * k
            for _ in range(2):
                mj.msm_bigint_batch(ck, sets, scalars_are_mont=True)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                mj.msm_bigint_batch(ck, sets, scalars_are_mont=True)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            res.append(round(sorted(ts)[2], 3))
        print("MZK_PRE_C", os.environ.get("MZK_PRE_C", "default"), "curve", cid, "n=2^%d" % ln, "one MSM ms", res[0], "batch of 5 ms", res[1], "shape", mlib.msm_last_shape(), flush=True)
        ck.release()
