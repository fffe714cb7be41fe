"""Radix2EvaluationDomain -- mirror of the ark-poly surface the jf-plonk prover drives:

    Radix2EvaluationDomain::new(n)                       plonk/src/proof_system/prover.rs:55
    GeneralEvaluationDomain::new(..).get_coset(g)        prover.rs:57-60, 545
    fft / fft_in_place      (coefficients -> evaluations) prover.rs:554-567
    ifft / ifft_in_place    (evaluations -> coefficients) prover.rs:672, relation/src/constraint_system.rs:1172-1257

Polynomials and evaluation vectors are (len, 4) uint64 arrays of Montgomery Fr limbs: numpy on the
host (copied to the GPU and back per call, as the Rust slices would be) or torch int64 CUDA tensors
(transformed in place in HBM, asynchronously on the current stream).  Every transform runs in
libmi355zk's HIP kernels; there is no host implementation here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .params import CurveParams, curve as _curve, fr_to_mont


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class Radix2EvaluationDomain:
    def __init__(self, curve, log_size: int, offset_mont: np.ndarray | None = None):
        self.curve: CurveParams = _curve(curve)
        if log_size > self.curve.two_adicity:
            raise ValueError("domain larger than the field's two-adicity")   # ark-poly: new() returns None
        self.log_size_of_group = log_size
        self.size = 1 << log_size
        self.offset_mont = None if offset_mont is None else np.ascontiguousarray(offset_mont, dtype=np.uint64).reshape(4)

    # ---- constructors ------------------------------------------------------------------------
    @classmethod
    def new(cls, curve, num_coeffs: int) -> "Radix2EvaluationDomain":
        """Smallest power-of-two domain holding num_coeffs coefficients (ark-poly rounds up)."""
        size = 1 if num_coeffs <= 1 else 1 << (num_coeffs - 1).bit_length()
        return cls(curve, size.bit_length() - 1)

    def get_coset(self, offset) -> "Radix2EvaluationDomain":
        """offset: Python int (canonical) or 4 Montgomery limbs; `Fr::GENERATOR` for the quotient domain."""
        if isinstance(offset, int):
            offset = fr_to_mont(self.curve, [offset])[0]
        return Radix2EvaluationDomain(self.curve, self.log_size_of_group, offset)

    def coset_offset_is_one(self) -> bool:
        return self.offset_mont is None

    # ---- transforms --------------------------------------------------------------------------
    def _offset_ptr(self):
        return None if self.offset_mont is None else C.c_void_p(self.offset_mont.ctypes.data)

    def _run_host(self, data: np.ndarray, inverse: bool) -> np.ndarray:
        a = np.ascontiguousarray(data, dtype=np.uint64)
        if a.ndim != 2 or a.shape[1] != 4:
            raise ValueError("expected an (n, 4) uint64 array of Fr limbs")
        if a.shape[0] > self.size:
            raise ValueError("input longer than the domain")
        buf = np.zeros((self.size, 4), dtype=np.uint64)
        buf[:a.shape[0]] = a
        L = _lib.ensure_init()
        _lib.check(L.mzk_ntt(self.curve.curve_id, C.c_void_p(buf.ctypes.data), a.shape[0], self.log_size_of_group,
                             int(inverse), self._offset_ptr()), "mzk_ntt")
        return buf

    def _run_dev(self, t, inverse: bool, in_len: int | None, stream=None):
        import torch
        if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("expected a contiguous int64 CUDA tensor")
        if t.dim() == 2:
            batch, stride = 1, self.size
            ok = t.shape == (self.size, 4)
        else:
            batch, stride = t.shape[0], self.size
            ok = t.dim() == 3 and t.shape[1:] == (self.size, 4)
        if not ok:
            raise ValueError("expected shape (size, 4) or (batch, size, 4)")
        st = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
        L = _lib.ensure_init()
        _lib.check(L.mzk_ntt_dev(self.curve.curve_id, t.data_ptr(), self.size if in_len is None else in_len,
                                 self.log_size_of_group, int(inverse), self._offset_ptr(), batch, stride, st), "mzk_ntt_dev")
        return t

    def fft(self, coeffs):
        """Evaluate on the (coset) domain: out[i] = p(offset * w^i).  Host arrays: returns a new
        (size,4) array, input zero-padded.  CUDA tensors must already have `size` rows."""
        if _is_torch(coeffs):
            return self._run_dev(coeffs.clone(), False, None)
        return self._run_host(coeffs, False)

    def ifft(self, evals):
        if _is_torch(evals):
            return self._run_dev(evals.clone(), True, None)
        return self._run_host(evals, True)

    def fft_in_place(self, t, in_len: int | None = None, stream=None):
        """CUDA tensor (size,4) or (batch,size,4), transformed in HBM; rows >= in_len count as zero."""
        return self._run_dev(t, False, in_len, stream)

    def ifft_in_place(self, t, in_len: int | None = None, stream=None):
        return self._run_dev(t, True, in_len, stream)
