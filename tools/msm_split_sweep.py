"""Bucket split of the latency-bound MSM sizes (csrc/msm.hip: every bucket over 2^log_split threads): one MSM, a pair and a batch of five
at 2^13 .. 2^18 pairs on the table path, for log_split forced to 0 .. 3 against the library's rule -- each setting in a child process
(the switch is read once).   python tools/msm_split_sweep.py [curve [settings, e.g. rule,1,2]]"""
import os, subprocess, sys
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else os.getcwd())
import numpy as np, torch
import mpc_jellyfish_amd as mj
from mpc_jellyfish_amd import lib as mlib
L = mlib.ensure_init()
cid = int(sys.argv[1])
c = mj.params.CURVES[cid]
for ln in (13, 14, 15, 16, 17, 18):
    n = 1 << ln
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
    s = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=3).view(np.int64)).cuda()
    out = []
    for k in (1, 2, 5, 6):
        f = (lambda: mj.kzg.msm_bigint(ck, s, scalars_are_mont=True)) if k == 1 else (lambda: mj.kzg.msm_bigint_batch(ck, [s] * k, scalars_are_mont=True))
        for _ in range(3): f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            for _ in range(8): f()
            best = min(best, (time.perf_counter() - t0) / 8 * 1e3)
        out.append("x%d %.3f" % (k, best))
    print("split", os.environ.get("MZK_MSM_FORCE_SPLIT", "rule"), "curve", cid, "log", ln, " ".join(out), flush=True)
    ck.release()
'''
cid = sys.argv[1] if len(sys.argv) > 1 else "0"
for sp in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("rule", "0", "1", "2", "3")):
    env = dict(os.environ)
    if sp != "rule": env["MZK_MSM_FORCE_SPLIT"] = sp
    subprocess.run([sys.executable, "-c", CHILD, cid], env=env, check=False)
