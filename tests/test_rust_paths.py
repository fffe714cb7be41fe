"""CPU suite: integration/rust/ (the Rust shim and the fixture generator -- source only, never compiled here) names only items
that exist in the reference tree (tools/check_rust_paths.py).  Skipped where the reference is absent (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rust_sources_name_existing_reference_items():
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference tree absent")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rust_paths.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
