import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
c = mj.params.BLS12_381
n = 1 << 20
fixed = mj.params.random_fr_mont(c, 18 * n, seed=31).reshape(18, n, 4)
wit = torch.from_numpy(mj.params.random_fr_mont(c, 7 * (n + 3), seed=32).reshape(7, n + 3, 4).view(np.int64)).cuda()
slab = torch.zeros((7, 8 * n, 4), dtype=torch.int64, device="cuda")
slab[:, :n + 3] = wit
ch = mj.plonk.Challenges(0x1234567, 0x89abcde, 0xf012345)
for classes in ([0], [0, 1], [0, 1, 2, 3], list(range(8))):
    pk = mj.plonk.ProvingKeyDevice.register(c, n, list(fixed[:13]), list(fixed[13:]), [1, 2, 3, 4, 5], classes=classes)
    out = mj.plonk.compute_quotient_chunked_dev(pk, ch, slab, n + 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        mj.plonk.compute_quotient_chunked_dev(pk, ch, slab, n + 3, out_dev=out)
    torch.cuda.synchronize()
    print("classes", len(classes), "chunked local ms", round((time.perf_counter() - t0) / 3 * 1e3, 3))
    pk.release()
r = torch.zeros((8, n, 4), dtype=torch.int64, device="cuda")
q = mj.plonk.combine_quotient_classes(c, n, r)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    mj.plonk.combine_quotient_classes(c, n, r, out_dev=q)
torch.cuda.synchronize()
print("combine ms", round((time.perf_counter() - t0) / 3 * 1e3, 3))
