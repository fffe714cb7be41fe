"""tools/ntt_time_one.py <log_n> [<log_n> ...] -- coset forward / inverse transform times of the given sizes, per pass (library timers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
from importlib import import_module
mlib = import_module("mpc-jellyfish_amd.lib")
L = mlib.ensure_init()
c = mj.params.CURVES[0]
for ln in [int(a) for a in sys.argv[1:]] or [22]:
    N = 1 << ln
    batch = max(1, (1 << 22) // N)
    x = torch.from_numpy(mj.params.random_fr_mont(c, N * batch, seed=1).view(np.int64)).cuda().reshape(batch, N, 4)
    d = mj.Radix2EvaluationDomain(c, ln).get_coset(c.fr_generator)
    for _ in range(3):
        d.fft_in_place(x); d.ifft_in_place(x)
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("fwd", d.fft_in_place), ("inv", d.ifft_in_place)):
        L.mzk_profile_reset(); L.mzk_profile_enable(1)
        for _ in range(10):
            fn(x)
        torch.cuda.synchronize()
        L.mzk_profile_enable(0)
        tot, cnt = mlib.profile_get("ntt_total")
        pas, pc = mlib.profile_get("ntt_pass")
        res[name] = (tot / cnt, pas / pc, pc // cnt)
    print("2^%d x %d: fwd %.4f ms (pass %.4f x %d)  inv %.4f ms (pass %.4f)" % (ln, batch, res["fwd"][0], res["fwd"][1], res["fwd"][2], res["inv"][0], res["inv"][1]), flush=True)
