import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
for cid in (0, 1):
    c = mj.params.CURVES[cid]
    for ln in (16, 18, 20, 22, 23, 24, 25):
        N = 1 << ln
        x = torch.from_numpy(mj.params.random_fr_mont(c, N, seed=1).view(np.int64)).cuda()
        d = mj.Radix2EvaluationDomain(c, ln).get_coset(c.fr_generator)
        d.fft_in_place(x); d.ifft_in_place(x)
        torch.cuda.synchronize()
        L.mzk_profile_reset(); L.mzk_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(4):
            d.fft_in_place(x)
        torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / 4 * 1e3
        t0 = time.perf_counter()
        for _ in range(4):
            d.ifft_in_place(x)
        torch.cuda.synchronize()
        inv = (time.perf_counter() - t0) / 4 * 1e3
        L.mzk_profile_enable(0)
        p = mlib.profile_get("ntt_pass")
        print("curve", cid, "2^%d" % ln, "fwd ms", round(fwd, 3), "inv ms", round(inv, 3), "passes", p[1] // 8, "ns/elem/pass", round(p[0] / p[1] * 1e6 / N, 3), flush=True)
        del x
