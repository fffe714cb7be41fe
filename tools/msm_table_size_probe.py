"""Does the accumulation's rate hold when the fixed-base table is tens of GB?  Table-path MSMs at 2^20 / 2^22 / 2^24 pairs on BLS12-381
(tables of 1.5 / 6.1 / 24.4 GB): accumulate ms per MSM and mixed additions per second.  python tools/msm_table_size_probe.py [logs..]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
for ln in [int(a) for a in sys.argv[1:]] or [20, 22, 24]:
    c = mj.params.CURVES[0]
    n = 1 << ln
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n - 1)
    s = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=3).view(np.int64)).cuda()
    for _ in range(2):
        mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
    L.mzk_profile_reset(); L.mzk_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
    L.mzk_profile_enable(0)
    acc, cnt = mlib.profile_get("msm_accumulate")
    srt, _ = mlib.profile_get("msm_sort")
    red, _ = mlib.profile_get("msm_reduce")
    cb, w, m = mlib.msm_last_shape()
    a = acc / max(cnt, 1)
    print("log", ln, "shape", (cb, w, m), "wall %.2f ms" % wall, "sort %.3f accumulate %.3f reduce %.3f" % (srt / 3, a, red / 3),
          "-> %.2f G mixed additions / s" % (n * w / a / 1e6), flush=True)
    ck.release()
    del s
    torch.cuda.empty_cache()
    L.mzk_workspace_release()
