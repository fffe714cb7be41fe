import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
n = 1 << 20
for cid, ultra in ((0, False), (1, True)):
    c = mj.params.CURVES[cid]
    W, nsel = (6, 14) if ultra else (5, 13)
    fixed = mj.params.random_fr_mont(c, (nsel + W + 4) * n, seed=31).reshape(nsel + W + 4, n, 4)
    tabs = {nm: fixed[nsel + W + i] for i, nm in enumerate(mj.plonk.PLOOKUP_TABLE_POLYS)} if ultra else None
    pk = mj.plonk.ProvingKeyDevice.register(c, n, list(fixed[:nsel]), list(fixed[nsel:nsel + W]), list(range(1, W + 1)), plookup=tabs)
    rows = W + 2 + (3 if ultra else 0)
    wit = torch.from_numpy(mj.params.random_fr_mont(c, rows * (n + 3), seed=32).reshape(rows, n + 3, 4).view(np.int64)).cuda()
    slab = torch.zeros((rows, 8 * n, 4), dtype=torch.int64, device="cuda")
    out = torch.empty((8 * n, 4), dtype=torch.int64, device="cuda")
    ch = mj.plonk.Challenges(0x1234567, 0x89abcde, 0xf012345, 0x777)
    def run():
        slab[:, :n + 3] = wit
        mj.plonk.compute_quotient_polynomial_dev(pk, ch, slab, n + 3, out)
    run(); torch.cuda.synchronize()
    L.mzk_profile_reset(); L.mzk_profile_enable(1)
    for _ in range(3):
        run()
    torch.cuda.synchronize(); L.mzk_profile_enable(0)
    k = mlib.profile_get("plonk_quotient_kernel"); t = mlib.profile_get("plonk_quotient_total")
    print("ultra" if ultra else "turbo", "kernel ms", round(k[0] / k[1], 3), "round ms", round(t[0] / t[1], 3), "checksum", int(out.sum().item()) & 0xffffffff)
    pk.release()
