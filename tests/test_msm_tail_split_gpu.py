"""The accumulation schedule of LARGE plain-path MSMs (csrc/msm.hip: whole-bucket threads for the first 7/8 of the ranks, the last eighth
split four ways -- round 5) against the oracle and against the schedules it replaced.

The plain path is what the reference's call site runs when nothing is precomputed (`msm_bigint`,
/root/reference/primitives/src/pcs/univariate_kzg/mod.rs:109-111) and what bench.py's headline times: BASELINE configs[1], 2^20 pairs on
BLS12-381 = 16 windows x 2^15 buckets.  Every case is pinned by the trapdoor identity commit(p) = [p(beta)]G computed with the C oracle
(SURVEY.md 8(c)(4)); the schedules -- the default, the rule of rounds 4-5 (MZK_MSM_NO_TAIL_SPLIT=1), other tail fractions and splits --
run in child processes (the switches are read once per process) and must print the same affine points."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import sys
import numpy as np
root = sys.argv[1]
for p in (root, root + "/oracle"):
    sys.path.insert(0, p)
import torch
import mpc_jellyfish_amd as mj
import cref
from importlib import import_module
lib = import_module("mpc-jellyfish_amd.lib")
lib.init(0)
L = mj.load()
L.mzk_msm_set_precompute(0)                       # the plain path: W windows x own bucket sets over the registered bases


def big(v):
    return np.array([[(v >> (64 * i)) & 0xffffffffffffffff for i in range(4)]], dtype=np.uint64)


for curve_id, log_n in ((0, 20), (1, 21)):
    n = 1 << log_n
    c = mj.params.CURVES[curve_id]
    beta = 0x3c4d5e6f708192a3b4c5d6e7f8091a2b3c4d5e6f708192a3b4c5d6e7f8091a2b % c.r
    pp = mj.UnivariateProverParam.gen_srs_for_testing(curve_id, beta, n - 1 + 12350)
    beta_m = mj.params.fr_to_mont(c, [beta])[0]
    dense = mj.params.random_fr_mont(c, n + 12345, seed=4100 + log_n)
    skew = dense[:n].copy()
    skew[::2] = cref.fr_convert(curve_id, big(c.r - 2), True)[0]            # every second scalar equal: over-long buckets in every window
    small = dense[:n].copy()                                                    # the last 3/4 small counters: the high windows' buckets thin out
    vals = np.zeros((n - n // 4, 4), dtype=np.uint64)
    vals[:, 0] = np.arange(n - n // 4, dtype=np.uint64)
    small[n // 4:] = cref.fr_convert(curve_id, vals, True)
    for name, sc, off, m in (("dense", dense, 0, n), ("ragged", dense, 5, n + 12345), ("short", dense, 3, n - 54321), ("skew", skew, 0, n), ("small_tail", small, 0, n)):
        s = np.ascontiguousarray(sc[:m])
        t = torch.from_numpy(s.view(np.int64)).cuda()
        L.mzk_profile_reset()
        L.mzk_profile_enable(1)
        jac = mj.msm_bigint(pp, t, scalars_are_mont=True, base_offset=off)
        L.mzk_profile_enable(0)
        comb = lib.profile_get("msm_split_combine")[1]
        c_bits, n_win, n_buckets = lib.msm_last_shape()
        aff = cref.jac_to_affine(curve_id, jac)[0]
        # trapdoor: sum_i s_i [beta^(off+i)]G = [beta^off p(beta)]G
        p_beta = cref.poly_eval(curve_id, s, beta_m)
        k = mj.params.limbs_to_int(cref.fr_convert(curve_id, p_beta.reshape(1, 4), False)[0]) * pow(beta, off, c.r) % c.r
        ok = bool(np.array_equal(aff, cref.g1_mul_gen(curve_id, k)))
        print("CASE", curve_id, log_n, name, n_win, n_buckets, int(comb > 0), int(ok), aff.tobytes().hex())
    pp.release()
"""


def _run(env_extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", _SCRIPT, root], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = {}
    for line in r.stdout.split("\n"):
        f = line.split()
        if f and f[0] == "CASE":
            out[(int(f[1]), f[3])] = dict(n_win=int(f[4]), n_buckets=int(f[5]), combined=int(f[6]), ok=int(f[7]), point=f[8])
    assert len(out) == 10, r.stdout[-2000:]
    return out


def test_plain_path_large_msm_schedules_agree_with_the_oracle(gpu):
    default = _run({})
    for key, v in default.items():
        assert v["ok"] == 1, ("default schedule differs from [p(beta)]G", key)
        assert v["n_win"] > 1 and v["n_win"] * v["n_buckets"] >= 1 << 18, ("not the several-set plain path", key, v)
        if key[1] in ("dense", "ragged"):                    # (fewer than 32 entries per bucket -- "short", skewed scalars -- : whole-bucket threads)
            assert v["combined"] == 1, ("the split path (and its combine launch) did not run", key)
    for switches in ({"MZK_MSM_NO_TAIL_SPLIT": "1"},                                          # every bucket halved (rounds 4-5)
                     {"MZK_MSM_FORCE_SPLIT": "0"},                                             # whole-bucket threads everywhere
                     {"MZK_MSM_TAIL_FRAC_LOG": "1", "MZK_MSM_TAIL_SPLIT": "3"},                # half of the ranks, eight ways
                     {"MZK_MSM_TAIL_FRAC_LOG": "6", "MZK_MSM_TAIL_SPLIT": "1"}):               # the last 1/64, halved
        other = _run(switches)
        for key, v in other.items():
            assert v["ok"] == 1, (switches, key)
            assert v["point"] == default[key]["point"], (switches, key)
        if "MZK_MSM_FORCE_SPLIT" in switches:
            assert all(v["combined"] == 0 for v in other.values())
