"""CPU: the MSM's host Horner tail (csrc/host_tail.hpp: the bucket sets of a plain-path MSM summed on a small pool of sleeping worker
threads, the calling thread running the chain of doublings) gives the same point as the serial order it replaces -- also with empty
bit-sums, with several callers at once (one gets the pool, the others fall back to the serial order) and without workers -- and the
pool is clean under ThreadSanitizer.  Host logic only: tools/host_tail_bench.cpp includes the product's host headers, no GPU, no library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "host_tail_bench.cpp")
INC = os.path.join(ROOT, "mpc-jellyfish_amd", "csrc")


def _build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-pthread", "-I", INC, "-o", exe, SRC] + flags)
    return exe


def test_pooled_tail_is_the_serial_point(tmp_path):
    exe = _build(tmp_path, "htb", ["-O2"])
    for workers in ("3", "0", "15"):
        out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, MZK_HOST_TAIL_THREADS=workers), timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = [ln for ln in out.stdout.splitlines() if "same point" in ln]
        assert len(lines) == 8 and all(ln.endswith("same point: yes") for ln in lines), out.stdout
        assert all("(pool of %s)" % workers in ln for ln in lines), out.stdout


@pytest.mark.parametrize("workers", ["3", "0"])
def test_concurrent_callers_under_thread_sanitizer(tmp_path, workers):
    exe = _build(tmp_path, "htb_tsan", ["-O1", "-g", "-fsanitize=thread"])
    out = subprocess.run([exe, "--stress"], capture_output=True, text=True, env=dict(os.environ, MZK_HOST_TAIL_THREADS=workers), timeout=600)
    assert out.returncode == 0 and "stress: ok" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr
