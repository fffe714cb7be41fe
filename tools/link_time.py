"""Time PlonkKzgSnark::link_proofs on the device at bench sizes (run on the GPU box):  python tools/link_time.py [log_n] [size]"""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_jellyfish_amd as mj

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
c = mj.params.BLS12_381
n = 1 << log_n
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 0x1234567, n + 2)
dom = mj.Radix2EvaluationDomain(c, log_n)
v1 = mj.params.random_fr_mont(c, n, seed=3)
v2 = mj.params.random_fr_mont(c, n, seed=4)
layout = mj.linking.GroupLayout(log_n - 2, 5, size)
start, _ = layout.range_in_nth_roots(log_n)
rows = start + 4 * np.arange(size)
v2[rows] = v1[rows]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
def masked(vals, seed):                                                      # a(X) + (b0 + b1 X)(X^n - 1): same values on H
    a = np.concatenate([dom.ifft(vals), np.zeros((2, 4), dtype=np.uint64)])
    t = dev(a)
    mj.poly.mask(c, [t], n, [[seed + 11, seed + 12]])
    return t


a1, a2 = masked(v1, 1), masked(v2, 2)
cm = lambda t: mj.UnivariateKzgPCS.commit(ck, t.cpu().numpy().view(np.uint64))
h1, h2 = mj.linking.LinkingHint(a1, cm(a1)), mj.linking.LinkingHint(a2, cm(a2))
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    diff, q = mj.linking.compute_linking_quotient(c, a1, a2, layout)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    link = mj.linking.link_proofs(h1, h2, layout, ck)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("log_n %d size %d: quotient %.2f ms, link_proofs %.2f ms" % (log_n, size, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
a2[7] = a1[3]                                                               # no longer a valid link: factor-by-factor path
torch.cuda.synchronize()
t0 = time.perf_counter()
diff, q = mj.linking.compute_linking_quotient(c, a1, a2, layout)
torch.cuda.synchronize()
print("with a remainder (factor by factor): quotient %.2f ms" % ((time.perf_counter() - t0) * 1e3))
