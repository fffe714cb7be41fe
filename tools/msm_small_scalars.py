"""MSM time by scalar SIZE: what committing in the Lagrange basis (scalars = witness values, often small) would cost against the
coefficient-form commitment (dense 255-bit scalars).  python tools/msm_small_scalars.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c = mj.params.CURVES[0]
n = 1 << lg
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
def limbs(vals):
    a = np.zeros((len(vals), 4), dtype=np.uint64)
    a[:, 0] = np.asarray(vals, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).cuda()
rng = np.random.default_rng(5)
cases = {
    "dense 255-bit": torch.from_numpy(mj.params.fr_bigints([int.from_bytes(rng.bytes(32), "little") % c.r for _ in range(1 << 12)] * (n >> 12)).view(np.int64)).cuda(),
    "0 .. n-1": limbs(np.arange(n)),
    "all ones": limbs(np.ones(n)),
    "random 8-bit": limbs(rng.integers(0, 256, n)),
    "random 32-bit": limbs(rng.integers(0, 1 << 32, n)),
    "random 64-bit": limbs(rng.integers(0, 1 << 63, n) * 2 + rng.integers(0, 2, n)),
    "booleans": limbs(rng.integers(0, 2, n)),
}
for name, s in cases.items():
    for _ in range(2):
        mj.kzg.msm_bigint(ck, s)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mj.kzg.msm_bigint(ck, s)
        ts.append((time.perf_counter() - t0) * 1e3)
    b = [s] * 5
    mj.kzg.msm_bigint_batch(ck, b)
    tb = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mj.kzg.msm_bigint_batch(ck, b)
        tb.append((time.perf_counter() - t0) * 1e3)
    print("2^%d %-14s single %.3f ms   batch of 5: %.3f ms" % (lg, name, sorted(ts)[3], sorted(tb)[2]), flush=True)
