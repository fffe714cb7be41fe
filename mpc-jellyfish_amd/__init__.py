"""mpc-jellyfish_amd -- host-side mirror of the jf-plonk prover's arithmetic boundary on MI355X.

The package is a thin Python layer over libmi355zk.so (HIP, gfx950; C ABI in include/mzk.h):

    params   curve / field constants and host-side encodings (Python ints <-> Montgomery limbs)
    lib      ctypes binding of the C ABI; raises if the HIP library is missing (no CPU fallback)
    domain   Radix2EvaluationDomain mirror  (ark-poly surface used at prover.rs:54-62,545-567,672)
    kzg      UnivariateKzgPCS mirror        (primitives/src/pcs/univariate_kzg/mod.rs:90-161, srs.rs)
    plonk    TurboPlonk quotient round      (plonk/src/proof_system/prover.rs:512-759)
    sharding multi-GPU split of the commit path

The directory name carries a hyphen, so import it with
    importlib.import_module("mpc-jellyfish_amd")      or      import mpc_jellyfish_amd   (shim at the repo root).
"""
from . import params  # noqa: F401
from .lib import MzkError, lib_path, load  # noqa: F401
from .domain import Radix2EvaluationDomain  # noqa: F401
from . import batch, linking, plonk, poly, prover, rng, sharding, snark, transcript  # noqa: F401
from .kzg import (Commitment, PCSError, UnivariateKzgPCS, UnivariateProverParam,  # noqa: F401
                  jacobian_to_affine, msm_bigint, msm_bigint_batch)
