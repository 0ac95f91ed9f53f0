"""BP+OSD decoder: host mirror of `BeliefPropagationOSDDecoder`
(src/decoders/belief_propagation_osd.jl:17-29, 49-61).

BP runs on the MI355X (the hot path); the ordered-statistics step is host code inside
libldpc_mi355x.so (`ldpc_osd_postprocess_batch`, bit-packed GF(2) elimination threaded over
the batch) -- BASELINE config 5 asks for exactly that split ("BP+OSD post-processing on host").
"""
from __future__ import annotations

import ctypes
from typing import Tuple

import numpy as np

from . import _capi
from .decoder import AbstractDecoder, BeliefPropagationDecoder, _pattern_of, syndrome_bytes


class OSDPostProcessor:
    """Owns the `ldpc_osd` handle: bit-packed H and the OSD order.  Needs no GPU."""

    def __init__(self, H, osd_order: int = 0):
        M = _pattern_of(H)
        self.s, self.n = int(M.shape[0]), int(M.shape[1])
        self.osd_order = int(osd_order)
        colptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(M.indices, dtype=np.int64)
        self._h = ctypes.c_void_p()
        _capi.check(_capi.lib().ldpc_osd_create(self.s, self.n, int(rowval.size), colptr.ctypes.data,
                                                rowval.ctypes.data, self.osd_order, ctypes.byref(self._h)))

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _capi.lib().ldpc_osd_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def postprocess(self, syn_bs, bp_err_bn, llr_bn, nthreads: int = 0) -> np.ndarray:
        """syn [B][s] u8 (0/1), bp_err [B][n] u8, llr [B][n] f64 -> errors [B][n] u8."""
        syn = np.ascontiguousarray(syn_bs, dtype=np.uint8)
        e = np.ascontiguousarray(bp_err_bn, dtype=np.uint8)
        L = np.ascontiguousarray(llr_bn, dtype=np.float64)
        B = syn.shape[0]
        assert syn.shape == (B, self.s) and e.shape == (B, self.n) and L.shape == (B, self.n)
        out = np.empty((B, self.n), dtype=np.uint8)
        _capi.check(_capi.lib().ldpc_osd_postprocess_batch(self._h, B, syn.ctypes.data, e.ctypes.data,
                                                           L.ctypes.data, out.ctypes.data, int(nthreads)))
        return out


class BeliefPropagationOSDDecoder(AbstractDecoder):
    """`BeliefPropagationOSDDecoder(H, per, max_iters; osd_order=0)` (belief_propagation_osd.jl:26-29)."""

    def __init__(self, H, per: float, max_iters: int, *, osd_order: int = 0, **bp_kwargs):
        # (exact LLRs: OSD orders the bits by reliability, :53-55 -- two that differ beyond the 21st bit must not tie)
        bp_kwargs.setdefault("llr_exact", True)
        self.bp_decoder = BeliefPropagationDecoder(H, per, max_iters, **bp_kwargs)   # :27
        self.H = H                                                                    # :21
        self.osd_order = int(osd_order)                                               # :23
        self._osd = OSDPostProcessor(H, osd_order)

    def decode_(self, syndrome) -> Tuple[np.ndarray, bool]:
        """`decode!(decoder::BeliefPropagationOSDDecoder, syndrome)` (:49-61): returns
        (error estimate as a Bool vector, whether BP converged)."""
        syn = syndrome_bytes(np.asarray(syndrome).reshape(-1))
        bp = self.bp_decoder
        if syn.size != bp.s:
            raise IndexError(f"syndrome has length {syn.size}, decoder has {bp.s} checks")
        err, conv, llr, _ = bp.decode_batch_host(syn.reshape(1, -1), want_llr=True)        # :51-52
        bp.scratch.err[:] = err[0]
        bp.scratch.log_probabs[:] = llr[0]
        out = self._osd.postprocess(syn.reshape(1, -1), err, llr, nthreads=1)              # :53-60
        return out[0].astype(np.bool_), bool(conv[0])

    def batchdecode_(self, syndromes, errors, success=None, nthreads: int = 0):
        """Batch form.  The reference takes the generic per-column loop for BP+OSD
        (abstract_decoder.jl:31-42, test_bposd_decoder.jl:49-57); one BP launch plus one threaded
        OSD pass gives the same columns."""
        syndromes = np.asarray(syndromes)
        B = syndromes.shape[1]
        if success is None:
            success = np.empty(B, dtype=np.bool_)
        assert syndromes.shape[1] == errors.shape[1]
        assert syndromes.shape[1] == len(success)
        bp = self.bp_decoder
        syn_bs = np.ascontiguousarray(syndrome_bytes(syndromes).T)
        err, conv, llr, _ = bp.decode_batch_host(syn_bs, want_llr=True)
        out = self._osd.postprocess(syn_bs, err, llr, nthreads=nthreads)
        errors[:, :] = out.T
        success[:] = conv.astype(np.bool_)
        if B > 0:
            bp.scratch.err[:] = err[-1]
            bp.scratch.log_probabs[:] = llr[-1]
        return errors, success

    def batchdecode_device(self, syn, nthreads: int = 0):
        """HBM-resident batch (BASELINE config 5 shape): `syn` is a [B][s] uint8 torch tensor on the
        GPU.  BP runs on the device; only the syndromes that still need OSD travel to the host:
        with osd_order = 0 a converged syndrome is returned unchanged by the reference's shortcut
        (belief_propagation_osd.jl:66-74: zero residual), so its OSD call is skipped; with
        osd_order > 0 every syndrome is post-processed, like the reference.
        Returns (errors [B][n] uint8 tensor, converged [B] uint8 tensor, number sent to OSD)."""
        import torch

        bp = self.bp_decoder
        B = int(syn.shape[0])
        dev = syn.device
        err = torch.empty((B, bp.n), dtype=torch.uint8, device=dev)
        conv = torch.empty(B, dtype=torch.uint8, device=dev)
        if self.osd_order == 0 and getattr(self, "_osd_frac", 0.0) <= 0.2:
            # Two passes: BP without LLRs for everybody (no 8n-byte LLR row per syndrome: 1.9 instead of
            # 2.4 ms per 2^20 BB-72 syndromes), then the few unconverged ones once more WITH LLRs -- the
            # decoder is deterministic per syndrome, so their hard decisions come out the same and the
            # LLRs are the ones the one-pass run would have written.  If a batch turns out to need OSD for
            # more than a fifth of its syndromes, later batches go back to one pass.
            bp.decode_batch_device(syn, err, conv, None, None)
            idx = torch.nonzero(conv == 0, as_tuple=False).flatten()
            k = int(idx.numel())
            self._osd_frac = k / max(B, 1)
            if k:
                sub = syn[idx].contiguous()
                e2 = torch.empty((k, bp.n), dtype=torch.uint8, device=dev)
                c2 = torch.empty(k, dtype=torch.uint8, device=dev)
                l2 = torch.empty((k, bp.n), dtype=torch.float64, device=dev)
                bp.decode_batch_device(sub, e2, c2, l2, None)
                out = self._osd.postprocess(sub.cpu().numpy(), e2.cpu().numpy(), l2.cpu().numpy(), nthreads=nthreads)
                err[idx] = torch.from_numpy(out).to(dev)
            return err, conv, k
        llr = torch.empty((B, bp.n), dtype=torch.float64, device=dev)
        bp.decode_batch_device(syn, err, conv, llr, None)
        if self.osd_order == 0:
            idx = torch.nonzero(conv == 0, as_tuple=False).flatten()
        else:
            idx = torch.arange(B, device=dev)
        k = int(idx.numel())
        self._osd_frac = k / max(B, 1) if self.osd_order == 0 else 1.0
        if k:
            out = self._osd.postprocess(syn[idx].cpu().numpy(), err[idx].cpu().numpy(), llr[idx].cpu().numpy(),
                                        nthreads=nthreads)
            err[idx] = torch.from_numpy(out).to(dev)
        return err, conv, k
