import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
cid, ln = int(sys.argv[1]), int(sys.argv[2])
c = mj.params.CURVES[cid]
N = 1 << ln
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, N + 2)
s = torch.from_numpy(mj.params.random_fr_mont(c, N + 3, seed=3).view(np.int64)).cuda()
for _ in range(3):
    mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)
torch.cuda.synchronize()
