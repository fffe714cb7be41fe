"""tools/ntt_two_pass_probe.py -- the 2^22-point transform as TWO passes of 2^11-point radices on 4096-element tiles (MZK_NTT_RADICES=11,11
MZK_NTT_TILE_LOG_RT=12, experiment switches of csrc/ntt.cuh / ntt.hip) against the shipping three passes: digest of the output and time per
transform, BN254 (the lazy-value bound of ntt_fx.cuh allows 22 stages without a reduction there: 8 + 66 < 128; BLS12-381 does not: 4 + 66 > 64).
Run once per configuration (the switches are read once per process):  python tools/ntt_two_pass_probe.py <label>"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpc_jellyfish_amd as mj
from importlib import import_module
mlib = import_module("mpc-jellyfish_amd.lib")
L = mlib.ensure_init()
c = mj.params.CURVES[1]
ln, N = 22, 1 << 22
x0 = mj.params.random_fr_mont(c, N, seed=1)
x = torch.from_numpy(x0.view(np.int64)).cuda()
d = mj.Radix2EvaluationDomain(c, ln)
t0 = time.time()
d.fft_in_place(x)
torch.cuda.synchronize()
first = time.time() - t0
dig = hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:16]
for _ in range(3):
    d.fft_in_place(x)
torch.cuda.synchronize()
L.mzk_profile_reset(); L.mzk_profile_enable(1)
t0 = time.perf_counter()
for _ in range(20):
    d.fft_in_place(x)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
L.mzk_profile_enable(0)
tot, cnt = mlib.profile_get("ntt_total")
pas, pc = mlib.profile_get("ntt_pass")
print("%s: BN254 2^22 plain forward: digest %s, %.4f ms per transform (library timer), %.4f ms wall, %d passes of %.4f ms; first call (plan build) %.2f s"
      % (sys.argv[1] if len(sys.argv) > 1 else "", dig, tot / cnt, wall, pc // cnt, pas / pc, first), flush=True)
