"""BP-OTS decoder: host mirror of `BPOTSDecoder` (src/decoders/bpots_decoder.jl:39-115) over the
ldpc_bpots_* entry points; decode!/batchdecode! as in :225-340 and abstract_decoder.jl:31-48."""
from __future__ import annotations

import ctypes
from typing import Tuple

import numpy as np

from . import _capi
from .decoder import AbstractDecoder, _pattern_of, syndrome_bytes


class BPOTSDecoder(AbstractDecoder):
    """`BPOTSDecoder(H, per::Float64, max_iters::Int; T::Int=9, C::Float64=2.0)`."""

    def __init__(self, H, per: float, max_iters: int, *, T: int = 9, C: float = 2.0, device=None, experiments=None):
        if not isinstance(per, float):
            raise TypeError("per must be a Float64")
        M = _pattern_of(H)
        self.per, self.max_iters, self.T, self.C = float(per), int(max_iters), int(T), float(C)
        self.s, self.n = int(M.shape[0]), int(M.shape[1])
        self.sparse_H = M
        colptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(M.indices, dtype=np.int64)
        self._h = ctypes.c_void_p()
        self._L = _capi.lib_for(experiments)
        _capi.check(self._L.ldpc_bpots_create(self.s, self.n, int(rowval.size), colptr.ctypes.data,
                                                  rowval.ctypes.data, self.per, self.max_iters, self.T, self.C,
                                                  -1 if device is None else int(device), ctypes.byref(self._h)), self._L)

    @property
    def kernel(self) -> int:
        """2 = LDS-resident kernel, 3 = node-parallel kernel with the messages in a global slot (ldpc_bpots_kernel)."""
        return int(self._L.ldpc_bpots_kernel(self._h))

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.ldpc_bpots_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_batch_host(self, syn_bs):
        """syn [B][s] uint8 -> (errors [B][n] u8 = best_decisions, converged [B] u8, iters [B] i32)."""
        syn = np.ascontiguousarray(syn_bs, dtype=np.uint8)
        B = int(syn.shape[0])
        if syn.ndim != 2 or syn.shape[1] != self.s:
            raise AssertionError("syndrome length does not match the number of checks")
        err = np.empty((B, self.n), dtype=np.uint8)
        conv = np.empty(B, dtype=np.uint8)
        its = np.empty(B, dtype=np.int32)
        _capi.check(self._L.ldpc_bpots_decode_batch(self._h, B, syn.ctypes.data, err.ctypes.data,
                                                    conv.ctypes.data, its.ctypes.data), self._L)
        return err, conv, its

    def decode_(self, syndrome) -> Tuple[np.ndarray, bool]:
        """`decode!(decoder::BPOTSDecoder, syndrome)`: (best_decisions as Int vector, converged)."""
        syn = syndrome_bytes(np.asarray(syndrome).reshape(-1))
        if syn.size != self.s:
            raise IndexError(f"syndrome has length {syn.size}, decoder has {self.s} checks")
        err, conv, _ = self.decode_batch_host(syn.reshape(1, -1))
        return err[0].astype(np.int64), bool(conv[0])

    def batchdecode_(self, syndromes, errors, success=None):
        syndromes = np.asarray(syndromes)
        B = syndromes.shape[1]
        if success is None:
            success = np.empty(B, dtype=np.bool_)
        assert syndromes.shape[1] == errors.shape[1]
        assert syndromes.shape[1] == len(success)
        err, conv, _ = self.decode_batch_host(np.ascontiguousarray(syndrome_bytes(syndromes).T))
        errors[:, :] = err.T
        success[:] = conv.astype(np.bool_)
        return errors, success
