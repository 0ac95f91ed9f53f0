"""ctypes front end of oracle/bpots_oracle.c plus the portable tanh/atanh it shares with the HIP
kernel (TEST INFRASTRUCTURE; parity unpinned -- see bpots_oracle.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def _lib(libm: bool = False):
    """libbpots_oracle.so (portable tanh / atanh, shared with the HIP kernel) or, with libm=True,
    libbpots_oracle_libm.so (the host's libm: shares nothing with the product)."""
    if libm not in _LIBS:
        name = "libbpots_oracle_libm.so" if libm else "libbpots_oracle.so"
        so = os.path.join(_HERE, name)
        srcs = [os.path.join(_HERE, "bpots_oracle.c"), os.path.join(_HERE, "Makefile"),
                os.path.join(_HERE, "..", "ldpcdecoders.jl_amd", "csrc", "portable_math.h")]
        if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
            subprocess.check_call(["make", "-s", "-C", _HERE, name])
        L = ctypes.CDLL(so)
        i64, f64, vp = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
        L.bpots_oracle_create.restype = vp
        L.bpots_oracle_create.argtypes = [i64, i64, i64, vp, vp, f64, i64, i64, f64]
        L.bpots_oracle_destroy.argtypes = [vp]
        L.bpots_oracle_decode_batch.argtypes = [vp, i64, vp, vp, vp, vp]
        _LIBS[libm] = L
    return _LIBS[libm]


class BPOTSOracle:
    """`BPOTSDecoder(H, per, max_iters; T, C)` + decode! on the CPU (bpots_decoder.jl:39-340)."""

    def __init__(self, csc, shape, per, max_iters, T=9, C=2.0, libm: bool = False):
        self._L = _lib(libm)
        self.s, self.n = int(shape[0]), int(shape[1])
        self.colptr = np.ascontiguousarray(csc[0], dtype=np.int64)
        self.rowval = np.ascontiguousarray(csc[1], dtype=np.int64)
        self._h = self._L.bpots_oracle_create(self.s, self.n, int(self.rowval.size), self.colptr.ctypes.data,
                                             self.rowval.ctypes.data, float(per), int(max_iters), int(T), float(C))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self, "_L", None) is not None:
            self._L.bpots_oracle_destroy(h)

    def batchdecode(self, syndromes):
        syn = np.ascontiguousarray(syndromes, dtype=np.uint8)
        B = syn.shape[0]
        err = np.zeros((B, self.n), dtype=np.uint8)
        conv = np.zeros(B, dtype=np.uint8)
        its = np.zeros(B, dtype=np.int32)
        self._L.bpots_oracle_decode_batch(self._h, B, syn.ctypes.data, err.ctypes.data, conv.ctypes.data, its.ctypes.data)
        return err, conv, its
