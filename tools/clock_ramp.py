"""tools/clock_ramp.py -- per-step time of the headline's MSM (2^20 pairs, BLS12-381, variable base) from a card that has idled:
the first steps after idle run slower than the sustained ones (what profiles/r05_f_warmup_ab.txt shows through bench.py's averages).
    python tools/clock_ramp.py [--steps 80] [--idle 3.0]
Prints step times in groups and, when rocm-smi answers an ordinary user, the shader clock before / during / after."""
import argparse
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_jellyfish_amd as mj                       # noqa: E402
from importlib import import_module                  # noqa: E402


def sclk():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20)
        return "; ".join(l.strip() for l in r.stdout.splitlines() if "sclk" in l.lower())[:200]
    except Exception as e:                           # noqa: BLE001
        return "rocm-smi: %r" % (e,)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--idle", type=float, default=3.0)
    a = ap.parse_args()
    mlib = import_module("mpc-jellyfish_amd.lib")
    L = mlib.init(0)
    curve = mj.params.BLS12_381
    n = 1 << 20
    pp = mj.UnivariateProverParam.gen_srs_for_testing(curve, 0x1234567 % curve.r, n - 1)
    d_scalars = torch.from_numpy(mj.params.random_fr_mont(curve, n, seed=7).view(np.int64)).to("cuda:0")
    L.mzk_msm_set_precompute(0)
    mj.msm_bigint(pp, d_scalars, scalars_are_mont=True)            # first call: workspace growth, not part of the picture
    torch.cuda.synchronize()
    for rep in range(3):
        time.sleep(a.idle)
        print("rep %d, after %.1f s idle: %s" % (rep, a.idle, sclk()), flush=True)
        ts = []
        for _ in range(a.steps):
            t0 = time.perf_counter()
            mj.msm_bigint(pp, d_scalars, scalars_are_mont=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("  after %d steps: %s" % (a.steps, sclk()))
        print("  steps 1-5  : " + " ".join("%.3f" % t for t in ts[:5]))
        for lo in range(5, a.steps, 15):
            g = ts[lo:lo + 15]
            print("  steps %2d-%2d: mean %.3f ms (min %.3f, max %.3f)" % (lo + 1, lo + len(g), sum(g) / len(g), min(g), max(g)))


if __name__ == "__main__":
    main()
