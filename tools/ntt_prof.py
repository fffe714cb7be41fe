#!/usr/bin/env python3
"""tools/ntt_prof.py -- an NTT-only workload for rocprofv3: `reps` transforms of 2^log_n points of each kind, in this order:
plain forward, coset forward, coset inverse, and the quotient round's shape (zero-padded coset forward to the internal form).
With --kernel-trace the dispatches of nttx_pass_kernel come in groups of `passes`; tools/ntt_prof_summary.py splits them by pass."""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mpc_jellyfish_amd as mj

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=22)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--curve", type=int, default=0)
args = ap.parse_args()
c = mj.params.CURVES[args.curve]
N = 1 << args.log_n
x = torch.from_numpy(mj.params.random_fr_mont(c, N, seed=1).view(np.int64)).cuda()
plain = mj.Radix2EvaluationDomain(c, args.log_n)
coset = plain.get_coset(c.fr_generator)
for _ in range(args.reps):
    plain.fft_in_place(x)
for _ in range(args.reps):
    coset.fft_in_place(x)
for _ in range(args.reps):
    coset.ifft_in_place(x)
torch.cuda.synchronize()
print("done")
