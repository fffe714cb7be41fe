"""Wall time of mzk_srs_lagrange_from_srs (the Lagrange-basis key from the points of an SRS: an inverse NTT over the group) by domain size.
    python tools/lagrange_key_time.py [max_log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpc_jellyfish_amd as mj
top = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for cid in (0, 1):
    c = mj.params.CURVES[cid]
    for lg in ([top] if top > 16 else range(10, top + 1, 2)):
        n = 1 << lg
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
        t0 = time.perf_counter()
        key = ck.lagrange_key(n)
        dt = time.perf_counter() - t0
        want = mj.UnivariateProverParam.gen_lagrange_srs_for_testing(c, 12345, n)
        ok = np.array_equal(key.powers_of_g(n - 4, 7), want.powers_of_g(n - 4, 7))
        print("%s 2^%d: %.2f s (%.1f us per butterfly), matches the trapdoor key: %s" % (c.name, lg, dt, dt * 1e6 / max(1, n // 2 * lg), ok), flush=True)
        for p in (ck, key, want):
            p.release()
