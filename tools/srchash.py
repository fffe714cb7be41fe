"""tools/srchash.py -- hash of the kernel sources a PMC pass was collected on.  tools/pmc_aggregate.py stores it in
profiles/*_traffic.json; bench.py recomputes it and reports `traffic` / `valu_issue` as null (with a note) when the kernels
have changed since the counters were read, instead of quoting stale numbers."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc-jellyfish_amd", "csrc")
# kernels AND what decides their launch geometry / pipeline (the .hip host sides, the generated constants, the build flags)
MSM_SOURCES = ("msm.cuh", "msm_pre.cuh", "ecx.cuh", "fx.cuh", "msm.hip", "constants.cuh", "Makefile")
NTT_SOURCES = ("ntt_fx.cuh", "ntt.cuh", "fs.cuh", "fx.cuh", "ntt.hip", "constants.cuh", "Makefile")


def sha16(files) -> str:
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print("msm", sha16(MSM_SOURCES), "ntt", sha16(NTT_SOURCES))
