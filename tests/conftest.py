import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def pyref():
    import pyref as P
    return P


@pytest.fixture(scope="session")
def cref():
    """The C oracle (oracle/cpu_ref.c), built on demand."""
    import cref as R
    R.lib()
    return R


@pytest.fixture(scope="session")
def mj():
    import mpc_jellyfish_amd as m
    return m


@pytest.fixture(scope="session")
def gpu(mj):
    """libmi355zk bound to cuda:0; fails loudly when the HIP extension or the GPU is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from importlib import import_module
    lib = import_module("mpc-jellyfish_amd.lib")
    lib.init(0)
    return lib


# ---- helpers shared by the tests -------------------------------------------------------------------
def fr_mont_limbs(c, ints):
    """canonical Python ints -> (n,4) uint64 Montgomery limbs (big-int arithmetic, test side)."""
    R = 1 << 256
    out = np.zeros((len(ints), 4), dtype=np.uint64)
    for i, v in enumerate(ints):
        m = v % c.r * R % c.r
        for j in range(4):
            out[i, j] = (m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_ints(a):
    a = np.asarray(a, dtype=np.uint64)
    a = a.reshape(-1, a.shape[-1])
    return [sum(int(a[i, j]) << (64 * j) for j in range(a.shape[1])) for i in range(a.shape[0])]


def fr_from_mont_limbs(c, a):
    rinv = pow(1 << 256, -1, c.r)
    return [v * rinv % c.r for v in limbs_ints(a)]


def affine_limbs(c, pts):
    """list of (x,y) canonical / None -> (n,2,fq_limbs) uint64 Montgomery; None -> (0,0)."""
    L = c.fq_limbs
    R = 1 << (64 * L)
    out = np.zeros((len(pts), 2, L), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is None:
            continue
        for k in range(2):
            m = p[k] * R % c.q
            for j in range(L):
                out[i, k, j] = (m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def affine_from_limbs(c, xy):
    """(2,fq_limbs) Montgomery -> (x,y) canonical or None."""
    v = limbs_ints(np.asarray(xy).reshape(2, c.fq_limbs))
    if v[0] == 0 and v[1] == 0:
        return None
    rinv = pow(1 << (64 * c.fq_limbs), -1, c.q)
    return (v[0] * rinv % c.q, v[1] * rinv % c.q)


def jacobian_to_affine_ints(c, xyz):
    """(3,fq_limbs) Montgomery Jacobian -> canonical affine (x,y) or None, by big-int arithmetic."""
    v = limbs_ints(np.asarray(xyz).reshape(3, c.fq_limbs))
    rinv = pow(1 << (64 * c.fq_limbs), -1, c.q)
    X, Y, Z = (t * rinv % c.q for t in v)
    if Z == 0:
        return None
    zi = pow(Z, -1, c.q)
    return (X * zi * zi % c.q, Y * zi * zi * zi % c.q)


def golden_pt(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


# general (non-bench) circuits: the builders live with the oracle (oracle/pyref_circuit.py) so that tests/golden/make_proof_golden.py, which
# imports nothing outside oracle/, proves the same circuits the GPU tests rebuild from the same seed
from pyref_circuit import general_circuit as build_circuit, general_ultra_circuit as build_ultra_circuit  # noqa: E402,F401


def verifying_key(mj, pc, pk, num_inputs):
    """VerifyingKey of preprocess (snark.rs:562-594) as the verifier restatement takes it."""
    pt = lambda cm: None if cm.is_infinity() else affine_from_limbs(pc, cm.xy)
    sel, sig = pk.vk_commitments()
    vk = {"domain_size": pk.n, "num_inputs": num_inputs, "k": list(pk.k), "selector_comms": [pt(x) for x in sel],
          "sigma_comms": [pt(x) for x in sig], "plookup": None}
    if pk.ultra:
        names = ("range_table_comm", "key_table_comm", "table_dom_sep_comm", "q_dom_sep_comm")
        vk["plookup"] = dict(zip(names, [pt(x) for x in pk.plookup_vk_commitments()]))
    return vk
