"""CPU: csrc/fs.cuh -- the constant-operand Barrett product on signed lazy limbs that every NTT butterfly runs -- compiled for the
HOST (the header is host/device code) and verified against Python big integers: plan-time records (w, floor(w 2^261 / p)),
congruence and range of fs_mulc on random, signed-extreme and edge multiplicands, limb class, fs_canonical.  The device build of
the same header is checked bit for bit against the Montgomery product in tools/fx_bench.hip (profiles/r02_fx_bench.txt) and, end
to end, by the NTT parity tests."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fs_mulc_and_canonical_against_big_integers(tmp_path):
    exe = str(tmp_path / "fs_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "mpc-jellyfish_amd", "csrc"),
                           os.path.join(ROOT, "tools", "fs_check.cpp"), "-o", exe])
    cases = subprocess.run([exe, "6000"], capture_output=True, check=True).stdout
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fs_check.py")], input=cases, capture_output=True)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert b"12000 cases ok" in out.stdout, out.stdout
