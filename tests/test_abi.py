"""CPU suite: the C-ABI library loads and exports every symbol include/mzk.h declares, the host
mirror's pure-host logic behaves like the reference's, and nothing computes without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mzk.h")).read()
    return sorted(set(re.findall(r"MZK_API\s+[\w\s\*]+?\b(mzk_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mj):
    L = mj.load()
    names = declared_symbols()
    assert len(names) >= 27
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/mzk.h but not exported by libmi355zk.so"
    from importlib import import_module
    lib = import_module("mpc-jellyfish_amd.lib")
    assert set(lib.EXPORTS) == set(names)
    assert L.mzk_version().startswith(b"libmi355zk")
    assert L.mzk_strerror(0) == b"ok" and L.mzk_strerror(-3) == b"no HIP device"


def test_no_cpu_fallback_without_gpu(mj):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    L = mj.load()
    assert L.mzk_init(-1) == -3                                # MZK_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.mzk_last_error()
    dom = mj.Radix2EvaluationDomain.new(0, 8)
    with pytest.raises(mj.MzkError):
        dom.fft(np.zeros((8, 4), dtype=np.uint64))
    with pytest.raises(mj.MzkError):
        mj.UnivariateProverParam.gen_srs_for_testing(0, 5, 4)


def test_domain_shapes_follow_the_reference(mj):
    P = mj.params
    # plonk/src/constants.rs:18-20 and SURVEY.md section 8: quotient domain is 8n for Turbo and Ultra
    for n, w in ((1 << 10, 5), (1 << 20, 5), (1 << 22, 6)):
        ratio = P.domain_size_ratio(n, w)
        assert ratio == (6 if w == 5 else 7)
        d = mj.Radix2EvaluationDomain.new(0 if w == 5 else 1, ratio * n)
        assert d.size == 8 * n
    d = mj.Radix2EvaluationDomain.new(0, 1000)
    assert d.size == 1024 and d.log_size_of_group == 10 and d.coset_offset_is_one()
    assert not d.get_coset(P.BLS12_381.fr_generator).coset_offset_is_one()
    assert mj.Radix2EvaluationDomain.new(1, 1).size == 1
    with pytest.raises(ValueError):
        mj.Radix2EvaluationDomain(1, 29)                       # BN254 two-adicity is 28


def test_host_encodings(mj, pyref):
    P = mj.params
    for c, pc in ((P.BLS12_381, pyref.BLS12_381), (P.BN254, pyref.BN254)):
        assert (c.r, c.q, c.fr_generator, c.two_adicity) == (pc.r, pc.q, pc.fr_gen, pc.two_adicity)
        vals = [0, 1, c.r - 1, 123456789 << 200]
        m = P.fr_to_mont(c, vals)
        assert P.fr_from_mont(c, m) == [v % c.r for v in vals]
        assert P.limbs_to_int(m[1]) == pyref.fr_to_mont(pc, 1)
        rnd = P.random_fr_mont(c, 5000, seed=3)
        assert rnd.shape == (5000, 4) and all(P.limbs_to_int(row) < c.r for row in rnd[:200])
        assert np.array_equal(rnd, P.random_fr_mont(c, 5000, seed=3))
        assert len({bytes(r) for r in rnd}) == 5000


def test_cpp_host_layer_builds_and_refuses_to_run_without_a_gpu(mj):
    """mpc-jellyfish_amd/host/ (C++ above the C ABI) compiles with g++ against include/mzk.h, and -- like everything else in
    the product -- has no CPU path: without a device it exits with the library's error."""
    import subprocess
    import torch
    host = os.path.join(ROOT, "mpc-jellyfish_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    binp = os.path.join(ROOT, "mpc-jellyfish_amd", "mzk_prove")
    assert os.path.exists(binp)
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = subprocess.run([binp, "0", "turbo", "32"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr
