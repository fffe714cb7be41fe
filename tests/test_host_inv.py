"""CPU: the host's inversion by division steps (csrc/hostinv.hpp: what h64::inv runs -- Jacobian -> affine at the end of every commit group,
the scalars of round 5) against the Fermat power it replaced, in all four fields (Fr and Fq of both curves): 0 -> 0, +-1, 2, the raw
images 2^k for EVERY bit position, p - 1, (p +- 1) / 2 and seeded random elements, each inverse multiplied back and the number of
30-step batches kept below the proven cap (tools/host_inv_bench.cpp, built with UBSan).  The proofs that go through it are compared byte
for byte with the oracle's in tests/test_golden_proofs_gpu.py."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_division_step_inverse_equals_fermat_inverse(tmp_path):
    exe = str(tmp_path / "host_inv_bench")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "mpc-jellyfish_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tools", "host_inv_bench.cpp")])
    out = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok") and "runtime error" not in out.stderr, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if "cases" in l]
    assert len(lines) == 4, out.stdout
    for name, cap in (("BLS12-381 Fr", 25), ("BN254 Fr", 25), ("BLS12-381 Fq", 37), ("BN254 Fq", 25)):
        line = next(l for l in lines if l.startswith(name + ":"))
        m = re.search(r"(\d+) cases, 0 mismatches; batches of 30 division steps: mean [\d.]+, max (\d+)", line)
        assert m and int(m.group(1)) > 3200 and int(m.group(2)) < cap, line
