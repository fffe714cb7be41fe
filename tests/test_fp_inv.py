"""CPU: the scalar-field inversion by Bernstein-Yang division steps (csrc/fp_inv.cuh, what the prover's batched divisions invert with since
round 5) against the Fermat power of fp.cuh on the host, both fields: 0 -> 0, +-1, 2, raw images at the limb boundaries of the 30-bit
form, p - 1, (p +- 1) / 2 and seeded random elements, each inverse multiplied back (tools/fp_inv_check.cpp; the same source the device
compiles).  The kernels that use it are compared with the oracle in tests/test_plonk_gpu.py and tests/test_ultra_gpu.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_division_step_inverse_equals_fermat_inverse(tmp_path):
    exe = str(tmp_path / "fp_inv_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "mpc-jellyfish_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tools", "fp_inv_check.cpp")])
    out = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok") and "runtime error" not in out.stderr, out.stdout + out.stderr
    assert "BLS12-381 Fr: 3020 cases, 0 mismatches" in out.stdout and "BN254 Fr: 3020 cases, 0 mismatches" in out.stdout
