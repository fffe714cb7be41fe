"""ctypes binding of libmi355zk.so (C ABI: include/mzk.h).  There is no CPU fallback: if the HIP
library is missing or no GPU is visible the calls raise."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MzkError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str):
        super().__init__(f"{where}: {detail} (code {code})")
        self.code = code


def lib_path() -> str:
    return os.environ.get("MZK_LIB_PATH") or os.path.join(_HERE, "libmi355zk.so")      # (MZK_LIB_PATH: A/B builds of the library, tools/ only)


_SIGS = {
    "mzk_init": [C.c_int32],
    "mzk_set_device": [C.c_int32],
    "mzk_get_device": [C.POINTER(C.c_int32)],
    "mzk_device_count": [C.POINTER(C.c_int32)],
    "mzk_dev_copy_peer": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p],
    "mzk_shutdown": [],
    "mzk_srs_register": [C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)],
    "mzk_srs_register_dev": [C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p],
    "mzk_srs_release": [C.c_uint64],
    "mzk_srs_generate_for_testing": [C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)],
    "mzk_srs_generate_for_testing_g": [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)],
    "mzk_srs_lagrange_from_srs": [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)],
    "mzk_srs_generate_lagrange_for_testing": [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)],
    "mzk_srs_slice": [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)],
    "mzk_srs_download": [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p],
    "mzk_srs_len": [C.c_uint64, C.POINTER(C.c_uint64)],
    "mzk_msm": [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p],
    "mzk_msm_dev": [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p],
    "mzk_msm_batch": [C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32, C.c_void_p],
    "mzk_msm_batch_dev": [C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32, C.c_void_p, C.c_void_p],
    "mzk_msm_affine": [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p],
    "mzk_g1_jacobian_to_affine": [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p],
    "mzk_g1_sum_jacobian": [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p],
    "mzk_ntt": [C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_void_p],
    "mzk_ntt_batch": [C.c_int32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32, C.c_int32, C.c_void_p],
    "mzk_ntt_dev": [C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p],
    "mzk_plonk_pk_register": [C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)],
    "mzk_plonk_pk_release": [C.c_uint64],
    "mzk_plonk_quotient_dev": [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plonk_quotient": [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_keccak_f1600": [C.c_void_p],
    "mzk_chacha_blocks": [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p],
    "mzk_plonk_pk_register_ultra": [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)],
    "mzk_plonk_quotient_ultra_dev": [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plonk_pk_register_chunked": [C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                                      C.POINTER(C.c_uint64)],
    "mzk_plonk_quotient_chunked_dev": [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plonk_quotient_chunked_flags_dev": [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p],
    "mzk_plonk_quotient_combine_dev": [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plonk_quotient_combine_classes_dev": [C.c_int32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plonk_quotient_top_dev": [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p],
    "mzk_plonk_quotient_combine_top_dev": [C.c_int32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p],
    "mzk_plookup_sorted_vec_dev": [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plookup_product_dev": [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plonk_perm_product_dev": [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plonk_perm_product": [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_plonk_gather_witness_dev": [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p],
    "mzk_poly_eval_dev": [C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_poly_lincomb_dev": [C.c_int32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p],
    "mzk_poly_split_quotient_dev": [C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p],
    "mzk_poly_mask_dev": [C.c_int32, C.c_uint32, C.POINTER(C.c_void_p), C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p],
    "mzk_poly_div_linear_dev": [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_poly_div_linear_rem_dev": [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_poly_degree_dev": [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p],
    "mzk_poly_div_roots_dev": [C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p],
    "mzk_dev_alloc": [C.c_uint64, C.POINTER(C.c_void_p)],
    "mzk_dev_free": [C.c_void_p],
    "mzk_dev_upload": [C.c_void_p, C.c_void_p, C.c_uint64],
    "mzk_dev_download": [C.c_void_p, C.c_void_p, C.c_uint64],
    "mzk_dev_sync": [],
    "mzk_stream_create": [C.POINTER(C.c_void_p)],
    "mzk_stream_destroy": [C.c_void_p],
    "mzk_stream_sync": [C.c_void_p],
    "mzk_stream_wait_stream": [C.c_void_p, C.c_void_p],
    "mzk_dev_upload_async": [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p],
    "mzk_dev_download_async": [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p],
    "mzk_dev_copy": [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p],
    "mzk_dev_copy2d": [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p],
    "mzk_dev_memset": [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p],
    "mzk_dev_memset2d": [C.c_void_p, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p],
    "mzk_profile_enable": [C.c_int32],
    "mzk_profile_get": [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)],
    "mzk_profile_reset": [],
    "mzk_host_alloc": [C.c_uint64, C.POINTER(C.c_void_p)],
    "mzk_host_free": [C.c_void_p],
    "mzk_host_register": [C.c_void_p, C.c_uint64],
    "mzk_host_unregister": [C.c_void_p],
    "mzk_msm_set_precompute": [C.c_int32],
    "mzk_srs_precompute": [C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_double)],
    "mzk_msm_last_shape": [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
    "mzk_srs_hbm_bytes": [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "mzk_plonk_pk_hbm_bytes": [C.c_uint64, C.POINTER(C.c_uint64)],
    "mzk_workspace_hbm_bytes": [C.POINTER(C.c_uint64)],
    "mzk_workspace_release": [],
    "mzk_launch_count": [C.POINTER(C.c_uint64)],
    "mzk_poly_eval_many_dev": [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    # the prover's rounds (csrc/prover.hip); mzk_comm* travels as a void pointer (sharding.TorchComm)
    "mzk_prover_create": [C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                          C.POINTER(C.c_uint64)],
    "mzk_prover_destroy": [C.c_uint64],
    "mzk_prover_vk_commitments": [C.c_uint64, C.c_void_p, C.c_void_p],
    "mzk_prover_set_wire_variables": [C.c_uint64, C.c_void_p, C.c_uint64],
    "mzk_prover_round1": [C.c_uint64, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p],
    "mzk_prover_round1_5": [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_prover_round2": [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_prover_round2_5": [C.c_uint64, C.c_void_p, C.c_void_p],
    "mzk_prover_round3": [C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p],
    "mzk_prover_round4": [C.c_uint64, C.c_void_p, C.c_void_p],
    "mzk_prover_round5": [C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.c_void_p],
    "mzk_prover_exchange_buffer": [C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)],
    "mzk_prover_set_peer_buffers": [C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)],
    "mzk_prover_poly_dev": [C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)],
    "mzk_prover_profile": [C.c_uint64, C.c_int32],
    "mzk_prover_timings": [C.c_uint64, C.c_char_p, C.c_uint64],
    "mzk_prover_hbm_bytes": [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
}
_STR_FUNCS = ("mzk_strerror", "mzk_last_error", "mzk_version")
EXPORTS = tuple(_SIGS) + _STR_FUNCS


def load():
    """dlopen the library (does not touch the GPU).  Raises FileNotFoundError if it was not built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(hipcc, gfx950). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  If this
        # library were loaded first it would bind /opt/rocm's copy, torch would then bring up its own, and the first one
        # would find no device.  Importing torch first makes both share torch's copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for name, args in _SIGS.items():
            f = getattr(L, name)
            f.argtypes = args
            f.restype = C.c_int32
        L.mzk_strerror.argtypes = [C.c_int32]
        for name in _STR_FUNCS:
            getattr(L, name).restype = C.c_char_p
        _LIB = L
    return _LIB


def check(rc: int, where: str):
    if rc != 0:
        L = load()
        raise MzkError(rc, where, f"{L.mzk_strerror(rc).decode()}: {L.mzk_last_error().decode()}")


_INIT_DEV = None


def init(device: int = -1):
    """mzk_init: bind this process to one GPU (idempotent)."""
    global _INIT_DEV
    L = load()
    check(L.mzk_init(device), "mzk_init")
    _INIT_DEV = device
    return L


def ensure_init():
    return load() if _INIT_DEV is not None else init(-1)


def profile_get(name: str):
    L = ensure_init()
    ms, cnt = C.c_double(), C.c_uint64()
    check(L.mzk_profile_get(name.encode(), C.byref(ms), C.byref(cnt)), "mzk_profile_get")
    return ms.value, cnt.value


def msm_last_shape():
    L = load()
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L.mzk_msm_last_shape(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value
