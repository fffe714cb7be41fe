"""Compatibility names: until round 4 `native.NativeProver` was the ctypes client of the round-level C ABI beside a Python sequencing of
the primitives in `prover.py`.  There is ONE implementation of the rounds now (csrc/prover.hip) and ONE client, `prover.TurboPlonkProver`;
this module re-exports it and the snark-level entry points under their old names."""
from .prover import (ERR_WRONG_QUOTIENT_DEGREE, WITNESS_DEV_VECTOR, WITNESS_DEV_WIRES, WITNESS_HOST_VECTOR, WITNESS_HOST_WIRES,  # noqa: F401
                     TurboPlonkProver as NativeProver, round3, round5)
from .snark import preprocess, prove  # noqa: F401
from .batch import batch_prove  # noqa: F401
