"""GPU parity: the HIP NTT (through the C ABI) against the committed definition-level vectors,
the C oracle on seeded inputs, and size-independent properties at the benchmark sizes."""
import numpy as np
import pytest

from conftest import fr_from_mont_limbs, fr_mont_limbs, load_golden

pytestmark = pytest.mark.gpu


def _dom(mj, c, log_n, offset):
    d = mj.Radix2EvaluationDomain(c.curve_id, log_n)
    return d if offset == 1 else d.get_coset(offset)


def test_ntt_matches_golden_vectors(gpu, mj, pyref):
    for case in load_golden("ntt_vectors"):
        c = pyref.CURVES[case["curve"]]
        log_n, offset = case["log_n"], int(case["offset"], 16)
        inp = fr_mont_limbs(c, [int(v, 16) for v in case["input"]])
        d = _dom(mj, c, log_n, offset)
        assert fr_from_mont_limbs(c, d.fft(inp)) == [int(v, 16) for v in case["forward"]], (c.name, log_n, len(inp), offset)
        assert fr_from_mont_limbs(c, d.ifft(inp)) == [int(v, 16) for v in case["inverse"]], (c.name, log_n, len(inp), offset, "inv")


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("log_n", [1, 5, 9, 10, 11, 13, 16, 17, 18, 20])
def test_ntt_matches_c_oracle(gpu, mj, cref, curve_id, log_n):
    """Seeded inputs, every pass-count regime (1, 2 and 3 passes), plain and coset, ragged inputs."""
    c = mj.params.CURVES[curve_id]
    n = 1 << log_n
    a = mj.params.random_fr_mont(c, n, seed=100 + log_n)
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    odd = mj.params.fr_to_mont(c, [0x123456789abcdef123])[0]
    for offset, off_limbs in ((1, None), (c.fr_generator, g), (0x123456789abcdef123, odd)):
        d = _dom(mj, c, log_n, offset)
        for in_len in (n, n // 8 + 3 if n >= 8 else n):
            x = a[:in_len]
            padded = np.zeros((n, 4), dtype=np.uint64)
            padded[:in_len] = x
            assert np.array_equal(d.fft(x), cref.ntt(curve_id, padded, log_n, False, off_limbs, threads=8)), (log_n, offset, in_len, "fwd")
            assert np.array_equal(d.ifft(x), cref.ntt(curve_id, padded, log_n, True, off_limbs, threads=8)), (log_n, offset, in_len, "inv")
        if log_n >= 17 and offset != 1:
            break                               # one coset is enough at the slow-oracle sizes


def test_ntt_edge_sizes(gpu, mj):
    c = mj.params.BLS12_381
    one = mj.params.fr_to_mont(c, [5])
    d0 = mj.Radix2EvaluationDomain(0, 0)
    assert np.array_equal(d0.fft(one), one) and np.array_equal(d0.ifft(one), one)
    assert np.array_equal(d0.get_coset(7).fft(one), one)
    d = mj.Radix2EvaluationDomain(0, 6)
    z = np.zeros((64, 4), dtype=np.uint64)
    assert not d.fft(z).any() and not d.ifft(z[:0]).any()
    # a size-1 transform is the identity of the ZERO-PADDED input: an empty input gives [0] whatever the staging buffer held before
    # (found by tools/soak.py in round 5)
    d.fft(mj.params.random_fr_mont(c, 64, seed=5))
    for dom in (d0, d0.get_coset(7)):
        for out in (dom.fft(one[:0]), dom.ifft(one[:0])):
            assert out.shape == (1, 4) and not out.any()
    const = d.fft(one)                          # constant polynomial evaluates to itself everywhere
    assert np.array_equal(const, np.repeat(one, 64, axis=0))
    with pytest.raises(ValueError):
        d.fft(np.zeros((65, 4), dtype=np.uint64))
    with pytest.raises(mj.MzkError):
        mj.lib.check(mj.load().mzk_ntt(7, z.ctypes.data, 64, 6, 0, None), "mzk_ntt")


def test_ntt_device_resident_batch_and_stream(gpu, mj, cref):
    """(batch, size, 4) CUDA tensor transformed in place on the current torch stream."""
    import torch
    c = mj.params.BN254
    log_n, n, batch = 12, 1 << 12, 5
    host = mj.params.random_fr_mont(c, batch * n, seed=7).reshape(batch, n, 4)
    t = torch.from_numpy(host.view(np.int64)).cuda()
    d = mj.Radix2EvaluationDomain(1, log_n).get_coset(c.fr_generator)
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d.fft_in_place(t, in_len=n // 2 + 1)
    s.synchronize()
    got = t.cpu().numpy().view(np.uint64)
    for b in range(batch):
        padded = host[b].copy()
        padded[n // 2 + 1:] = 0
        assert np.array_equal(got[b], cref.ntt(1, padded, log_n, False, g, threads=4)), b
    d.ifft_in_place(t)
    torch.cuda.synchronize()
    back = t.cpu().numpy().view(np.uint64)
    for b in range(batch):
        padded = host[b].copy()
        padded[n // 2 + 1:] = 0
        assert np.array_equal(back[b], padded)


@pytest.mark.parametrize("curve_id,log_n", [(0, 22), (0, 23), (1, 25), (0, 27)])
def test_ntt_full_size_properties(gpu, mj, cref, curve_id, log_n):
    """BASELINE sizes (C3: 2^22; C4 quotient domain 2^23; C5 quotient domain 2^25, BN254) and the largest supported transform (2^27, three passes of 9-bit radices, 4 GiB per vector):
    inverse(forward(x)) == x bit-exactly on the Fr::GENERATOR coset, linearity, and spot
    evaluations against Horner's rule on the oracle."""
    import torch
    c = mj.params.CURVES[curve_id]
    n = 1 << log_n
    x = mj.params.random_fr_mont(c, n, seed=log_n)
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    d = mj.Radix2EvaluationDomain(curve_id, log_n).get_coset(c.fr_generator)
    t = torch.from_numpy(x.view(np.int64)).cuda()
    d.fft_in_place(t)
    ev = t.cpu().numpy().view(np.uint64)
    if log_n <= 23:
        for i in (0, 1, 12345, n // 2 + 7, n - 1):
            pt = cref.domain_element(curve_id, log_n, i, g)
            assert np.array_equal(ev[i], cref.poly_eval(curve_id, x, pt)), i
    d.ifft_in_place(t)
    assert np.array_equal(t.cpu().numpy().view(np.uint64), x)
    # linearity on a sparse second input: NTT(x + y) - NTT(x) == NTT(y), y = e_k  => column of powers
    k = 5
    y = np.zeros((n, 4), dtype=np.uint64)
    y[k] = mj.params.fr_to_mont(c, [1])[0]
    ty = torch.from_numpy(y.view(np.int64)).cuda()
    d.fft_in_place(ty)
    evy = ty.cpu().numpy().view(np.uint64)
    for i in (0, 3, n - 2):
        pt = cref.domain_element(curve_id, log_n, i, g)             # (g w^i)
        pk = pt
        for _ in range(k - 1):
            pk = cref.fr_mul(curve_id, pk.reshape(1, 4), pt.reshape(1, 4))[0]
        assert np.array_equal(evy[i], pk), i


@pytest.mark.parametrize("curve_id", [0, 1])
def test_ntt_random_ragged_sweep(gpu, mj, cref, curve_id):
    """Random (size, input length, direction, coset) transforms -- every power of two up to 2^15 and input lengths that cut the
    first pass's skipped stages at arbitrary places (the zero-padding shortcut of the quotient round) -- and all-zero /
    single-coefficient inputs, against the C restatement of ark-poly's radix-2 transform."""
    import random
    c = mj.params.CURVES[curve_id]
    rng = random.Random(909 + curve_id)
    for case in range(48):
        log_n = 1 + case % 15
        n = 1 << log_n
        in_len = rng.choice([1, 2, n // 2, n // 2 + 1, n - 1, n, rng.randrange(1, n + 1), max(1, n >> rng.randrange(1, log_n + 1))])
        in_len = min(max(in_len, 1), n)
        offset = rng.choice([1, c.fr_generator, rng.randrange(2, c.r)])
        off_limbs = None if offset == 1 else mj.params.fr_to_mont(c, [offset])[0]
        a = mj.params.random_fr_mont(c, in_len, seed=7000 + case)
        if case % 7 == 3:
            a[:] = 0
        if case % 7 == 5:
            a[:] = 0
            a[in_len - 1] = mj.params.fr_to_mont(c, [rng.randrange(c.r)])[0]
        padded = np.zeros((n, 4), dtype=np.uint64)
        padded[:in_len] = a
        d = _dom(mj, c, log_n, offset)
        inverse = bool(case & 1)
        got = d.ifft(a) if inverse else d.fft(a)
        assert np.array_equal(got, cref.ntt(curve_id, padded, log_n, inverse, off_limbs, threads=4)), (case, log_n, in_len, offset, inverse)


# ---- the kernel configurations C3 / C4 / C5 actually run: transforms of >= 2^21 points (nttx_pass_kernel<., true>) --------------------
def _large_cases(mj, c, log_n):
    """(offset, offset limbs, input length): plain, Fr::GENERATOR coset (the quotient domain's), ragged zero-padded input."""
    n = 1 << log_n
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    return [(1, None, n), (c.fr_generator, g, n), (c.fr_generator, g, n // 8 + 3), (1, None, n // 2 + 1)]


@pytest.mark.parametrize("curve_id", [0, 1])
@pytest.mark.parametrize("log_n", [21, 22])
def test_ntt_large_matches_c_oracle_full_vector(gpu, mj, cref, curve_id, log_n):
    """Full-vector equality with the C restatement of ark-poly's radix-2 transform (oracle/ntt_impl.inc) on the stage-pair kernel the
    >= 2^21-point transforms take (csrc/ntt.hip: r4) -- forward and inverse, plain and coset, ragged input.  The reference's own check
    pattern for this call site is relation/src/constraint_system.rs:2028-2034 (iFFT of the evaluations against the polynomial)."""
    import os
    c = mj.params.CURVES[curve_id]
    n = 1 << log_n
    th = min(16, os.cpu_count() or 1)
    a = mj.params.random_fr_mont(c, n, seed=300 + log_n)
    for offset, off_limbs, in_len in _large_cases(mj, c, log_n):
        d = _dom(mj, c, log_n, offset)
        padded = np.zeros((n, 4), dtype=np.uint64)
        padded[:in_len] = a[:in_len]
        assert np.array_equal(d.fft(a[:in_len]), cref.ntt(curve_id, padded, log_n, False, off_limbs, threads=th)), (log_n, offset, in_len, "fwd")
        assert np.array_equal(d.ifft(a[:in_len]), cref.ntt(curve_id, padded, log_n, True, off_limbs, threads=th)), (log_n, offset, in_len, "inv")


_NTT_DIGEST_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import mpc_jellyfish_amd as mj
from importlib import import_module
import_module("mpc-jellyfish_amd.lib").init(0)
for curve_id in (0, 1):
    c = mj.params.CURVES[curve_id]
    for log_n in (21, 22):
        n = 1 << log_n
        a = mj.params.random_fr_mont(c, n, seed=300 + log_n)
        for offset, in_len in ((1, n), (c.fr_generator, n), (c.fr_generator, n // 8 + 3), (1, n // 2 + 1)):
            d = mj.Radix2EvaluationDomain(curve_id, log_n)
            d = d if offset == 1 else d.get_coset(offset)
            for inv in (False, True):
                out = d.ifft(a[:in_len]) if inv else d.fft(a[:in_len])
                print(curve_id, log_n, offset == 1, in_len, inv, hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest())
"""


def test_ntt_two_kernels_one_answer(gpu, mj):
    """The stage-pair kernel (default from 2^21 points) and the radix-2 form (MZK_NTT_NO_RADIX4=1, a child process: the switch is read
    once per process) give the same bytes on every case of the full-vector test above -- which pins both to the oracle."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MZK_NTT_NO_RADIX4="1")
    r = subprocess.run([sys.executable, "-c", _NTT_DIGEST_SCRIPT, root], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    want = {}
    for line in r.stdout.split("\n"):
        f = line.split()
        if len(f) == 6:
            want[(int(f[0]), int(f[1]), f[2] == "True", int(f[3]), f[4] == "True")] = f[5]
    assert len(want) == 2 * 2 * 4 * 2
    for curve_id in (0, 1):
        c = mj.params.CURVES[curve_id]
        for log_n in (21, 22):
            n = 1 << log_n
            a = mj.params.random_fr_mont(c, n, seed=300 + log_n)
            for offset, _, in_len in _large_cases(mj, c, log_n):
                d = _dom(mj, c, log_n, offset)
                for inv in (False, True):
                    out = d.ifft(a[:in_len]) if inv else d.fft(a[:in_len])
                    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == want[(curve_id, log_n, offset == 1, in_len, inv)], \
                        (curve_id, log_n, offset, in_len, inv)


def test_ntt_2p25_bn254_matches_c_oracle_full_vector(gpu, mj, cref):
    """C5's quotient domain (UltraPlonk / BN254, 2^22 gates -> 2^25 points; three stage-pair passes): the whole 2^25-point coset
    transform and its inverse against the C oracle (1 GiB per vector; the oracle takes 10-40 s on the box's host threads)."""
    import os
    import torch
    c = mj.params.BN254
    log_n, n = 25, 1 << 25
    th = min(16, os.cpu_count() or 1)
    x = mj.params.random_fr_mont(c, n, seed=2500)
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    d = mj.Radix2EvaluationDomain(1, log_n).get_coset(c.fr_generator)
    t = torch.from_numpy(x.view(np.int64)).cuda()
    d.fft_in_place(t)
    want = cref.ntt(1, x, log_n, False, g, threads=th)
    assert np.array_equal(t.cpu().numpy().view(np.uint64), want)
    del want
    t.copy_(torch.from_numpy(x.view(np.int64)))
    d.ifft_in_place(t)
    want = cref.ntt(1, x, log_n, True, g, threads=th)
    assert np.array_equal(t.cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("curve_id,log_n", [(1, 25), (0, 27)])
def test_ntt_largest_sizes_seeded_spot_evaluations(gpu, mj, cref, curve_id, log_n):
    """>= 60 seeded output indices of the forward coset transform against Horner's rule on the oracle
    (n field products per index, run on the host's threads), then inverse(forward(x)) == x on the whole vector."""
    import os
    import random
    from concurrent.futures import ThreadPoolExecutor
    import torch
    c = mj.params.CURVES[curve_id]
    n = 1 << log_n
    x = mj.params.random_fr_mont(c, n, seed=9000 + log_n)
    g = mj.params.fr_to_mont(c, [c.fr_generator])[0]
    d = mj.Radix2EvaluationDomain(curve_id, log_n).get_coset(c.fr_generator)
    t = torch.from_numpy(x.view(np.int64)).cuda()
    d.fft_in_place(t)
    ev = t.cpu().numpy().view(np.uint64)
    rng = random.Random(4242 + log_n)
    idx = sorted({0, 1, n // 2, n - 1} | {rng.randrange(n) for _ in range(60)})
    assert len(idx) >= 60

    def one(i):
        pt = cref.domain_element(curve_id, log_n, i, g)
        return i, np.array_equal(ev[i], cref.poly_eval(curve_id, x, pt))

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:       # ctypes releases the GIL: the Horner loops run in parallel
        bad = [i for i, ok in ex.map(one, idx) if not ok]
    assert not bad, bad
    d.ifft_in_place(t)
    assert np.array_equal(t.cpu().numpy().view(np.uint64), x)
