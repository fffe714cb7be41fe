"""batch_commit of 5 polynomials of 2^log_n coefficients (one mzk_msm_batch_dev): python tools/commit5_time.py [log_n] [curve_id]
MZK_MSM_NO_OVERLAP=1 disables the second sort stream (csrc/msm.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c = mj.params.CURVES[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
n = 1 << lg
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
polys = [torch.from_numpy(mj.params.random_fr_mont(c, n, seed=40 + i).view(np.int64)).cuda() for i in range(5)]
one = mj.jacobian_to_affine(c, np.stack([mj.kzg.msm_bigint(ck, p, scalars_are_mont=True) for p in polys]))
for _ in range(3):
    got = mj.kzg.msm_bigint_batch(ck, polys, scalars_are_mont=True)
assert np.array_equal(mj.jacobian_to_affine(c, got), one), "batch differs from the single MSMs"
ts = []
for _ in range(9):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mj.kzg.msm_bigint_batch(ck, polys, scalars_are_mont=True)
    ts.append((time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter()
for p in polys:
    mj.kzg.msm_bigint(ck, p, scalars_are_mont=True)
single = (time.perf_counter() - t0) * 1e3
print("%s 2^%d: batch of 5: median %.3f ms (min %.3f); five single MSMs %.3f ms; overlap %s" % (c.name, lg, sorted(ts)[4], min(ts), single, "off" if os.environ.get("MZK_MSM_NO_OVERLAP") else "on"))
