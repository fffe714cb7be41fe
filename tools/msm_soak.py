"""Soak of the batched MSM path (second sort stream, three buffer sets): many batches of ragged sizes, every result compared with the
single-MSM path.  python tools/msm_soak.py [iterations] [curve_id]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
c = mj.params.CURVES[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
rs = np.random.default_rng(7)
N = 1 << 18
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 4242, N + 2)
pool = torch.from_numpy(mj.params.random_fr_mont(c, N + 3, seed=11).view(np.int64)).cuda()
single = {}
bad = 0
t0 = time.time()
for it in range(iters):
    k = int(rs.integers(2, 9))
    lens = [int(rs.choice([1, 17, 1 << 10, 3000, 1 << 14, 50000, 1 << 16, 200000, N + 3])) for _ in range(k)]
    offs = [int(rs.integers(0, N + 3 - ln + 1)) for ln in lens]
    polys = [pool[o:o + ln] for o, ln in zip(offs, lens)]
    got = mj.jacobian_to_affine(c, mj.kzg.msm_bigint_batch(ck, polys, scalars_are_mont=True))
    for i, (o, ln) in enumerate(zip(offs, lens)):
        key = (o, ln)
        if key not in single:
            single[key] = mj.jacobian_to_affine(c, mj.kzg.msm_bigint(ck, pool[o:o + ln], scalars_are_mont=True)[None])[0]
        if not np.array_equal(got[i], single[key]):
            bad += 1
            print("MISMATCH iteration", it, "poly", i, key, flush=True)
    if it % 50 == 49:
        print("iteration", it + 1, "bad", bad, "%.1f s" % (time.time() - t0), flush=True)
print("soak done:", iters, "batches,", bad, "mismatches")
sys.exit(1 if bad else 0)
