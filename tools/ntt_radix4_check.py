"""One library build (MZK_LIB_PATH): SHA-256 of a 2^22 / 2^18 / 2^13 coset NTT of seeded data (all builds must print the same digests),
then wall time per transform at 2^16 .. 2^24 (MZK_NTT_NO_RADIX4=1: the round-3 form) (forward + inverse coset, 8 repetitions) and the ntt_pass event time."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
c = mj.params.CURVES[0]
dig = []
for ln in (22, 18, 13, 9):
    x = torch.from_numpy(mj.params.random_fr_mont(c, 1 << ln, seed=ln).view(np.int64)).cuda()
    d = mj.Radix2EvaluationDomain(c, ln).get_coset(c.fr_generator)
    d.fft_in_place(x)
    h = hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:16]
    d.ifft_in_place(x)
    ok = bool((x.cpu().numpy().view(np.uint64) == mj.params.random_fr_mont(c, 1 << ln, seed=ln)).all())
    dig.append("2^%d %s roundtrip %s" % (ln, h, ok))
print("; ".join(dig))
for ln in (16, 18, 20, 22, 24):
    N = 1 << ln
    x = torch.from_numpy(mj.params.random_fr_mont(c, N, seed=1).view(np.int64)).cuda()
    d = mj.Radix2EvaluationDomain(c, ln).get_coset(c.fr_generator)
    d.fft_in_place(x); d.ifft_in_place(x)
    torch.cuda.synchronize()
    L.mzk_profile_reset(); L.mzk_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(8):
        d.fft_in_place(x); d.ifft_in_place(x)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 16 * 1e3
    L.mzk_profile_enable(0)
    p = mlib.profile_get("ntt_pass")
    print("2^%d  %.4f ms per transform, pass %.4f ms" % (ln, per, p[0] / p[1]), flush=True)
    del x
