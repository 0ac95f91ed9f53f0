"""Host-side mirror of the reference's decoder interface for the BP hot path.

Julia is not available in this image, so the host layer above the C ABI is
written in Python with the reference's names, argument meaning and error
behaviour (Julia's trailing ``!`` is spelled with a trailing underscore):

    BeliefPropagationDecoder(H, per, max_iters)   src/decoders/belief_propagation.jl:38-67
    reset_(decoder)            -> decoder          :83-91
    decode_(decoder, syndrome) -> (err, converged) :121-188
    batchdecode_(decoder, syndromes, errors[, success]) -> (errors, success)
                                                   :220-231, src/decoders/abstract_decoder.jl:31-48

All arithmetic happens in the HIP kernels behind ``libldpc_mi355x.so``; this
file only marshals arrays.  Matrices keep the reference's orientation:
``syndromes`` is ``s x B`` and ``errors`` is ``n x B`` (one column per sample).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from . import _capi


class AbstractDecoder:
    """`abstract type AbstractDecoder` (src/decoders/abstract_decoder.jl:9).

    A concrete decoder implements ``decode_(self, syndrome) -> (guess, converged)``.
    """

    def decode_(self, syndrome):  # pragma: no cover - interface
        raise NotImplementedError


class BeliefPropagationScratchSpace:
    """Host mirror of the fields other code reads (belief_propagation.jl:3-18):
    ``log_probabs`` (read by BP+OSD, belief_propagation_osd.jl:52) and ``err``
    (returned by alias from decode!, :187).  The message matrices themselves
    live in HBM in the layout described in DESIGN.md and are not exposed."""

    def __init__(self, n: int, s: int, per: float):
        self.log_probabs = np.zeros(n, dtype=np.float64)
        self.channel_probs = np.full(n, per, dtype=np.float64)
        self.err = np.zeros(n, dtype=np.float64)


def _pattern_of(H) -> sp.csc_matrix:
    """`sparse(H)` (belief_propagation.jl:63): CSC pattern, rows ascending per column."""
    if sp.issparse(H):
        M = sp.csc_matrix(H, copy=True)
        M.sum_duplicates()
        M.eliminate_zeros()
    else:
        A = np.asarray(H)
        if A.ndim != 2:
            raise ValueError("H must be a matrix")
        M = sp.csc_matrix(A != 0)
    M.sort_indices()
    return M


def syndrome_bytes(x) -> np.ndarray:
    """Map syndrome entries to the ABI's uint8 alphabet.

    0/1 (Bool, Int or integral Float) pass through.  Any other integer keeps
    its parity -- ``(-1)^x`` at belief_propagation.jl:136 only depends on it --
    and is encoded as 2 + parity so that it can never satisfy the ``==`` of the
    convergence test (:181), exactly like an Int 2 or 3 in the reference.
    A non-integral float is a DomainError in Julia; here a ValueError.
    """
    a = np.asarray(x)
    if a.dtype == np.bool_:
        return a.astype(np.uint8)
    if a.dtype == np.uint8 and (a.size == 0 or a.max() <= 1):
        return a
    if np.issubdtype(a.dtype, np.floating):
        if not np.all(np.isfinite(a)) or np.any(a != np.rint(a)):
            raise ValueError("syndrome entries must be integral ((-1)^x is a DomainError otherwise)")
        a = a.astype(np.int64)
    elif not np.issubdtype(a.dtype, np.integer):
        raise TypeError(f"unsupported syndrome element type {a.dtype}")
    a64 = a.astype(np.int64)
    out = np.where((a64 == 0) | (a64 == 1), a64, 2 + (a64 & 1))
    return out.astype(np.uint8)


class BeliefPropagationDecoder(AbstractDecoder):
    """Drop-in for `BeliefPropagationDecoder(H, per::Float64, max_iters::Int)`.

    Extra keyword arguments select the device and tuning knobs; defaults give
    the reference behaviour (``llr_exact=True``: LLRs from the full posterior odds instead of their upper 32 bits;
    include/ldpc_mi355x.h ldpc_bp_options).  ``devices=[0, 1, ...]`` makes ``batchdecode_`` / ``decode_batch_host`` /
    ``decode_batch_device`` partition the batch over those GPUs from this one process
    (ldpc_bp_create_multi: contiguous shards, one handle and stream per device; ``exchange`` picks how the
    root-device form moves shards: 0 auto, 1 hipMemcpyPeer, 2 RCCL).  ``experiments=True`` binds the build of
    the library that reads the environment knobs (default: only when one of them is set).
    """

    def __init__(self, H, per: float, max_iters: int, *, device: Optional[int] = None,
                 devices: Optional[Sequence[int]] = None, exchange: int = 0,
                 waves_per_tile: int = 0, resident_tiles: int = 0, kernel_variant: int = 0,
                 defer_threshold: int = 0, llr_exact: bool = False, experiments: Optional[bool] = None):
        if not isinstance(per, float):
            raise TypeError("per must be a Float64 (reference signature: per::Float64)")
        if isinstance(max_iters, bool) or not isinstance(max_iters, (int, np.integer)):
            raise TypeError("max_iters must be an Int (reference signature: max_iters::Int)")
        M = _pattern_of(H)
        self.per = float(per)
        self.max_iters = int(max_iters)
        self.s, self.n = int(M.shape[0]), int(M.shape[1])
        self.sparse_H = M                      # columns = bits      (:63)
        self.sparse_HT = sp.csc_matrix(M.T)    # columns = checks    (:64)
        self.sparse_HT.sort_indices()
        self.scratch = BeliefPropagationScratchSpace(self.n, self.s, self.per)
        self._colptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
        self._rowval = np.ascontiguousarray(M.indices, dtype=np.int64)
        opts = _capi.BPOptions()
        opts.device = -1 if device is None else int(device)
        opts.waves_per_tile = int(waves_per_tile)
        opts.resident_tiles = int(resident_tiles)
        opts.kernel_variant = int(kernel_variant)   # 0 auto, 1 HBM-streaming, 2 LDS-resident, 3 node-parallel, 4 team
        opts.defer_threshold = int(defer_threshold)  # 0 auto (16), -1 off: straggler hand-off of the streaming kernel
        opts.llr_exact = 1 if llr_exact else 0       # LLRs from the full posterior odds (default: their upper 32 bits, within 5e-7)
        self._h = ctypes.c_void_p()
        self._m = None                         # the multi-device handle when devices= is given
        self._L = L = _capi.lib_for(experiments)
        if devices is not None:
            if device is not None:
                raise TypeError("give device= or devices=, not both")
            devs = (ctypes.c_int32 * len(devices))(*[int(x) for x in devices])
            m = ctypes.c_void_p()
            self._check(L.ldpc_bp_create_multi(len(devices), devs, int(exchange), self.s, self.n, int(self._rowval.size),
                                               self._colptr.ctypes.data, self._rowval.ctypes.data, self.per, self.max_iters,
                                               ctypes.byref(opts), ctypes.byref(m)))
            self._m = m
            self._h = ctypes.c_void_p(L.ldpc_bp_multi_handle(m, 0))   # info / timing: the root's handle
            self.devices = [int(x) for x in devices]
        else:
            self._check(L.ldpc_bp_create(self.s, self.n, int(self._rowval.size), self._colptr.ctypes.data,
                                         self._rowval.ctypes.data, self.per, self.max_iters,
                                         ctypes.byref(opts), ctypes.byref(self._h)))
            self.devices = None

    def _check(self, status: int) -> None:
        _capi.check(status, self._L)

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        m, self._m = getattr(self, "_m", None), None
        if m:
            self._L.ldpc_bp_destroy_multi(m)
        elif h:
            self._L.ldpc_bp_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection ------------------------------------------------------
    def info(self) -> _capi.BPInfo:
        info = _capi.BPInfo()
        self._check(self._L.ldpc_bp_get_info(self._h, ctypes.byref(info)))
        return info

    def multi_info(self) -> _capi.BPMultiInfo:
        """devices, exchange kind and the scatter / decode / gather times of the most recent root-device call
        (ldpc_bp_multi_get_info); only with devices=."""
        assert self._m, "not a multi-device decoder"
        info = _capi.BPMultiInfo()
        self._check(self._L.ldpc_bp_multi_get_info(self._m, ctypes.byref(info)))
        return info

    def last_status(self) -> None:
        """Wait for everything enqueued on the handle and raise LdpcError if a team of workgroups lost a
        member in one of those calls (ldpc_bp_last_status): their outputs must then be decoded again."""
        if self._m:
            self._check(self._L.ldpc_bp_multi_last_status(self._m))
        else:
            self._check(self._L.ldpc_bp_last_status(self._h))

    def last_timing(self, calls_back: int = 0) -> Tuple[float, float, int]:
        """(sweep_ms, total_ms, sum_iters) of a recent batch call (0 = the latest), from HIP
        events recorded on the stream the kernels ran on."""
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        self._check(self._L.ldpc_bp_call_timing(self._h, calls_back, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def phase_ticks(self, calls_back: int = 0) -> Tuple[int, int, int]:
        """Diagnostics: 100 MHz ticks in (check sweep, variable sweep, convergence test), summed over workgroups."""
        t = (ctypes.c_uint64 * 3)()
        self._check(self._L.ldpc_bp_call_phase_ticks(self._h, calls_back, ctypes.byref(t)))
        return int(t[0]), int(t[1]), int(t[2])

    # -- raw ABI calls ------------------------------------------------------
    def decode_batch_host(self, syn_bs: np.ndarray, want_llr: bool = False, want_iters: bool = False, out=None):
        """syn_bs: [B][s] uint8 C-contiguous.  Returns (errors [B][n] u8, converged [B] u8, llr|None, iters|None).
        `out` = (errors, converged) preallocated C-contiguous uint8 arrays to write into (a caller that
        reuses its buffers, like Julia's batchdecode!, avoids fresh-page faults on every call)."""
        syn_bs = np.ascontiguousarray(syn_bs, dtype=np.uint8)
        B = int(syn_bs.shape[0])
        if syn_bs.ndim != 2 or syn_bs.shape[1] != self.s:
            raise AssertionError("syndrome length does not match the number of checks")
        if out is not None:
            err, conv = out
            assert err.dtype == np.uint8 and err.flags.c_contiguous and err.shape == (B, self.n)
            assert conv.dtype == np.uint8 and conv.flags.c_contiguous and conv.shape == (B,)
        else:
            err = np.empty((B, self.n), dtype=np.uint8)   # every byte is written by the library
            conv = np.empty(B, dtype=np.uint8)
        llr = np.empty((B, self.n), dtype=np.float64) if want_llr else None
        its = np.empty(B, dtype=np.int32) if want_iters else None
        entry, handle = (self._L.ldpc_bp_decode_batch_multi, self._m) if self._m else (self._L.ldpc_bp_decode_batch, self._h)
        self._check(entry(handle, B, syn_bs.ctypes.data, err.ctypes.data, conv.ctypes.data,
                          llr.ctypes.data if want_llr else None, its.ctypes.data if want_iters else None))
        return err, conv, llr, its

    def decode_batch_device(self, syn, err, conv, llr=None, iters=None, stream: Optional[int] = None) -> None:
        """HBM-resident batch: torch CUDA(HIP) tensors, syn [B][s] u8, err [B][n] u8, conv [B] u8,
        llr [B][n] f64 | None, iters [B] i32 | None, all contiguous.  Asynchronous on `stream`
        (a hipStream_t as int; default = torch's current stream)."""
        import torch

        B = int(syn.shape[0])
        assert syn.is_cuda and err.is_cuda and conv.is_cuda
        assert syn.dtype == torch.uint8 and err.dtype == torch.uint8 and conv.dtype == torch.uint8
        assert syn.is_contiguous() and err.is_contiguous() and conv.is_contiguous()
        assert tuple(syn.shape) == (B, self.s) and tuple(err.shape) == (B, self.n) and conv.numel() == B
        if llr is not None:
            assert llr.is_cuda and llr.dtype == torch.float64 and llr.is_contiguous() and tuple(llr.shape) == (B, self.n)
        if iters is not None:
            assert iters.is_cuda and iters.dtype == torch.int32 and iters.is_contiguous() and iters.numel() == B
        if stream is None:
            stream = torch.cuda.current_stream(syn.device).cuda_stream
        entry, handle = ((self._L.ldpc_bp_decode_batch_multi_device, self._m) if self._m
                         else (self._L.ldpc_bp_decode_batch_device, self._h))   # multi: the tensors live on devices[0]
        self._check(entry(handle, B, syn.data_ptr(), err.data_ptr(), conv.data_ptr(),
                          llr.data_ptr() if llr is not None else None,
                          iters.data_ptr() if iters is not None else None, ctypes.c_void_p(stream)))

    # -- reference interface (methods; free functions below) ----------------
    def decode_(self, syndrome):
        return decode_(self, syndrome)


def reset_(decoder: BeliefPropagationDecoder) -> BeliefPropagationDecoder:
    """`reset!(bp_decoder)` (belief_propagation.jl:83-91).  Device scratch is reset
    inside every decode call, so only the host mirrors need clearing."""
    sc = decoder.scratch
    sc.log_probabs[:] = 0.0
    sc.channel_probs[:] = decoder.per
    sc.err[:] = 0.0
    return decoder


def decode_(decoder: BeliefPropagationDecoder, syndrome) -> Tuple[np.ndarray, bool]:
    """`decode!(decoder, syndrome)` (belief_propagation.jl:121-188).

    Returns ``(decoder.scratch.err, converged)``; like the reference the first
    element is the scratch vector itself (Float64 0.0/1.0), overwritten by the
    next call (:187)."""
    syn = syndrome_bytes(np.asarray(syndrome).reshape(-1))
    if syn.size != decoder.s:
        raise IndexError(f"syndrome has length {syn.size}, decoder has {decoder.s} checks")  # BoundsError
    reset_(decoder)
    err, conv, llr, _ = decoder.decode_batch_host(syn.reshape(1, -1), want_llr=True)
    decoder.scratch.err[:] = err[0]
    decoder.scratch.log_probabs[:] = llr[0]
    return decoder.scratch.err, bool(conv[0])


def batchdecode_(decoder: AbstractDecoder, syndromes, errors, success=None):
    """`batchdecode!(decoder, syndromes, errors[, success])`.

    For a BeliefPropagationDecoder the whole batch is one device call
    (belief_propagation.jl:220-231); any other AbstractDecoder takes the generic
    per-column loop (abstract_decoder.jl:31-42).  ``syndromes`` is ``s x B``,
    ``errors`` is ``n x B`` and is overwritten; ``success`` (length B, bool) is
    allocated when omitted (abstract_decoder.jl:44-48).  Returns
    ``(errors, success)``."""
    syndromes = np.asarray(syndromes) if not isinstance(syndromes, np.ndarray) else syndromes
    if syndromes.ndim != 2 or errors.ndim != 2:
        raise TypeError("syndromes and errors must be matrices")
    B = syndromes.shape[1]
    if success is None:
        success = np.empty(B, dtype=np.bool_)                     # Vector{Bool}(undef, B)
    assert syndromes.shape[1] == errors.shape[1]                  # :221
    assert syndromes.shape[1] == len(success)                     # :222
    if hasattr(decoder, "batchdecode_") and not isinstance(decoder, BeliefPropagationDecoder):
        return decoder.batchdecode_(syndromes, errors, success)   # a decoder with its own batch strategy
    if not isinstance(decoder, BeliefPropagationDecoder):
        for i in range(B):                                        # abstract_decoder.jl:35-39
            guess, conv = decoder.decode_(syndromes[:, i])
            success[i] = conv
            errors[:, i] = guess
        return errors, success
    if syndromes.shape[0] != decoder.s or errors.shape[0] != decoder.n:
        raise IndexError("syndromes/errors row count does not match the decoder")
    syn_bs = syndrome_bytes(syndromes).T                          # [B][s]; a view for column-major input
    err, conv, _, _ = decoder.decode_batch_host(syn_bs)
    errors[:, :] = err.T                                          # 0/1 -> eltype(errors)  (:227)
    success[:] = conv.astype(np.bool_)                            # :226
    if B > 0:
        # the reference leaves the scratch holding the last column's result; the LLRs of
        # the whole batch are not shipped back for that, the last column is re-decoded alone
        _, _, llr, _ = decoder.decode_batch_host(syn_bs[-1:], want_llr=True)
        decoder.scratch.err[:] = err[-1]
        decoder.scratch.log_probabs[:] = llr[0]
    return errors, success
