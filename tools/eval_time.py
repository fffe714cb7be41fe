import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
for cid in (0, 1):
    c = mj.params.CURVES[cid]
    n = 1 << 20
    t = torch.from_numpy(mj.params.random_fr_mont(c, 11 * (n + 3), seed=3).reshape(11, n + 3, 4).view(np.int64)).cuda()
    for batch, name in ((t[:6], "6 rows"), (t[6], "1 row"), (t[[8, 9, 10, 3, 4]], "gathered 5"), (t[:4], "4 rows"), (t[:3], "3 rows")):
        mj.poly.evaluate(c, batch, 12345)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            mj.poly.evaluate(c, batch, 12345)
        torch.cuda.synchronize()
        print(c.name if hasattr(c, 'name') else cid, name, round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
    t0 = time.perf_counter()
    g = t[[8, 9, 10, 3, 4]]
    torch.cuda.synchronize()
    print("gather", round((time.perf_counter() - t0) * 1e3, 3))
