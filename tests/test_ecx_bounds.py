"""CPU: the value / limb bounds of the reduced-radix EC formulas (csrc/ecx.cuh) for the pad sets that ship (tools/ecx_bounds.py): BLS12-381 Fq on
14 limbs with the wide pads, BN254 Fq on 9 limbs (7 bits of head-room) with the tight ones -- and the checker refuses 9 limbs with wide pads."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ecx_bounds_hold_for_the_shipped_pad_sets():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ecx_bounds.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BLS12-381" in out.stdout and "BN254" in out.stdout and "refused, as it must be" in out.stdout


def test_the_checked_pad_set_is_the_generated_one():
    """constants.cuh (generated) carries the limb count, invariant and pads that tools/ecx_bounds.py checks"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ecx_bounds as E
    text = open(os.path.join(ROOT, "mpc-jellyfish_amd", "csrc", "constants.cuh")).read()
    for name, struct in (("BLS12-381", "BlsFqX"), ("BN254", "BnFqX")):
        f = next(x for x in E.FIELDS if x.name == name)
        body = text[text.index("struct %s " % struct):]
        body = body[:body.index("\n};")]
        assert int(re.search(r"XN = (\d+);", body).group(1)) == f.xn
        assert int(re.search(r"XKXY = (\d+);", body).group(1)) == f.kxy
        for arr, k in (("XSUB_XY", f.pad_xy), ("XSUB_PQ", f.pad_pq), ("XSUB_2S", f.pad_2s)):
            limbs = [int(v, 16) for v in re.search(arr + r"\[\d+\] = \{([^}]*)\}", body).group(1).replace("u", "").split(",")]
            assert sum(v << (29 * i) for i, v in enumerate(limbs)) == k * f.p, (name, arr)
