"""ctypes front end of oracle/bp_oracle.c (TEST INFRASTRUCTURE; parity unpinned).

Follows the reference's call shapes so that tests read like the reference's:
``BPOracle(H, per, max_iters).decode(syndrome) -> (err, converged)`` mirrors
``decode!`` (belief_propagation.jl:121-188) and ``batchdecode`` mirrors
``batchdecode!`` (:220-231).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[ctypes.CDLL] = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (no GPU, no reference sources needed)."""
    so = os.path.join(_HERE, "libbp_oracle.so")
    src = os.path.join(_HERE, "bp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libbp_oracle.so"])
    return so


def _lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i64, f64, vp = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
        L.bp_oracle_create.restype = vp
        L.bp_oracle_create.argtypes = [i64, i64, i64, vp, vp, f64, i64, ctypes.c_int]
        L.bp_oracle_destroy.argtypes = [vp]
        L.bp_oracle_reset.argtypes = [vp]
        L.bp_oracle_decode.restype = ctypes.c_int
        L.bp_oracle_decode.argtypes = [vp, vp]
        L.bp_oracle_decode_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp]
        L.bp_oracle_err.restype = ctypes.POINTER(f64)
        L.bp_oracle_err.argtypes = [vp]
        L.bp_oracle_log_probabs.restype = ctypes.POINTER(f64)
        L.bp_oracle_log_probabs.argtypes = [vp]
        L.bp_oracle_last_iters.restype = i64
        L.bp_oracle_last_iters.argtypes = [vp]
        L.bp_oracle_messages.argtypes = [vp, vp, vp]
        _LIB = L
    return _LIB


def csc_from_dense(H) -> Tuple[np.ndarray, np.ndarray]:
    """0-based CSC pattern (colptr, rowval) of a dense 0/1 matrix, rows ascending
    inside each column -- what `sparse(H)` builds at belief_propagation.jl:63."""
    H = np.asarray(H)
    s, n = H.shape
    nzj, nzi = np.nonzero(H.T)  # column-major order: by column, rows ascending
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, nzj + 1, 1)
    colptr = np.cumsum(colptr).astype(np.int64)
    return colptr, nzi.astype(np.int64)


class BPOracle:
    def __init__(self, H=None, per: float = 0.0, max_iters: int = 0, *, csc=None, shape=None,
                 dense: bool = False):
        if csc is None:
            H = np.asarray(H)
            shape = H.shape
            csc = csc_from_dense(H)
        self.s, self.n = int(shape[0]), int(shape[1])
        self.colptr = np.ascontiguousarray(csc[0], dtype=np.int64)
        self.rowval = np.ascontiguousarray(csc[1], dtype=np.int64)
        self.nnz = int(self.rowval.size)
        self.per, self.max_iters = float(per), int(max_iters)
        self._h = _lib().bp_oracle_create(self.s, self.n, self.nnz, self.colptr.ctypes.data,
                                          self.rowval.ctypes.data, self.per, self.max_iters,
                                          1 if dense else 0)
        if not self._h:
            raise ValueError("bp_oracle_create rejected the CSC pattern")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _LIB is not None:
            _LIB.bp_oracle_destroy(h)

    def decode(self, syndrome) -> Tuple[np.ndarray, bool]:
        syn = np.ascontiguousarray(syndrome, dtype=np.uint8)
        assert syn.shape == (self.s,)
        conv = _lib().bp_oracle_decode(self._h, syn.ctypes.data)
        err = np.ctypeslib.as_array(_lib().bp_oracle_err(self._h), shape=(max(self.n, 1),))[: self.n].copy()
        return err, bool(conv)

    @property
    def log_probabs(self) -> np.ndarray:
        return np.ctypeslib.as_array(_lib().bp_oracle_log_probabs(self._h), shape=(max(self.n, 1),))[: self.n].copy()

    @property
    def last_iters(self) -> int:
        return int(_lib().bp_oracle_last_iters(self._h))

    def messages(self) -> Tuple[np.ndarray, np.ndarray]:
        b = np.zeros(max(self.nnz, 1)); c = np.zeros(max(self.nnz, 1))
        _lib().bp_oracle_messages(self._h, b.ctypes.data, c.ctypes.data)
        return b[: self.nnz], c[: self.nnz]

    def batchdecode(self, syndromes, want_llr: bool = True):
        """syndromes: [B][s] uint8 (row b = column b of the reference's s x B matrix).
        Returns errors [B][n] u8, converged [B] u8, llr [B][n] f64 | None, iters [B] i32."""
        syn = np.ascontiguousarray(syndromes, dtype=np.uint8)
        B = syn.shape[0]
        assert syn.shape == (B, self.s)
        errors = np.zeros((B, self.n), dtype=np.uint8)
        conv = np.zeros(B, dtype=np.uint8)
        llr = np.zeros((B, self.n), dtype=np.float64) if want_llr else None
        iters = np.zeros(B, dtype=np.int32)
        _lib().bp_oracle_decode_batch(self._h, B, syn.ctypes.data, errors.ctypes.data, conv.ctypes.data,
                                      llr.ctypes.data if want_llr else None, iters.ctypes.data)
        return errors, conv, llr, iters


_OSD_LIB = None


def osd_oracle_postprocess(H, syndrome, bp_err, log_probabs, osd_order: int) -> np.ndarray:
    """oracle/osd_oracle.c: belief_propagation_osd.jl:52-60 + osd (:63-125 / :127-209) for ONE syndrome.
    H dense m x n 0/1; returns the error estimate (n bytes)."""
    global _OSD_LIB
    if _OSD_LIB is None:
        so = os.path.join(_HERE, "libosd_oracle.so")
        src = os.path.join(_HERE, "osd_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libosd_oracle.so"])
        _OSD_LIB = ctypes.CDLL(so)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        _OSD_LIB.osd_oracle_postprocess.argtypes = [vp, i64, i64, vp, vp, vp, i64, vp]
    Hd = np.ascontiguousarray(np.asarray(H) != 0, dtype=np.uint8)
    m, n = Hd.shape
    syn = np.ascontiguousarray(syndrome, dtype=np.uint8)
    e = np.ascontiguousarray(bp_err, dtype=np.uint8)
    L = np.ascontiguousarray(log_probabs, dtype=np.float64)
    out = np.zeros(n, dtype=np.uint8)
    _OSD_LIB.osd_oracle_postprocess(Hd.ctypes.data, m, n, syn.ctypes.data, e.ctypes.data, L.ctypes.data,
                                    int(osd_order), out.ctypes.data)
    return out
