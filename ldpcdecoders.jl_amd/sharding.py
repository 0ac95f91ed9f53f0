"""batchdecode! across the GPUs of one node: one process per GPU, contiguous batch shards.

The columns of `syndromes` are decoded independently (belief_propagation.jl:224-228),
so the path shards with NO collective inside the decode: the root holds the caller's
`s x B` matrix (BASELINE config 4: one caller-held batch), hands every rank a contiguous
run of columns, each rank decodes its run with its own decoder handle, and the hard
decisions / flags / iteration counts travel back.  Exchange = grouped point-to-point
send/recv (``torch.distributed.batch_isend_irecv``; on the ``nccl`` backend that is an
RCCL ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI, each peer on its own
direct link to the root; on ``gloo`` the same code runs on CPU tensors -- the tests, and
the one-GPU rehearsal where device tensors are staged through the host).

World size 1 (or no process group at all) is the degenerate case of the same code: no
exchange, the root's shard is the whole batch and is decoded straight into the result
arrays.

When every rank already owns its syndromes (production ingest from pinned host memory per
GPU) nothing here is needed: ranks simply call the decoder on their shard.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous shards: rank g gets columns [g*B/G, (g+1)*B/G)."""
    return [((batch * g) // world, (batch * (g + 1)) // world) for g in range(world)]


def _default_device(group) -> torch.device:
    """Where the exchanged tensors live unless the caller says otherwise: the current HIP device under
    RCCL (an ``nccl`` broadcast of a CPU tensor fails), the host under gloo."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class _Clock:
    """Phase stamps of one call: HIP events on the current stream for device tensors (the exchange and the
    decode are asynchronous), host time otherwise."""

    def __init__(self, device: torch.device):
        self.cuda = device.type == "cuda"
        self.device = device
        self.marks = []

    def mark(self):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            self.marks.append(e)
        else:
            self.marks.append(time.perf_counter())

    def spans_ms(self) -> List[float]:
        if self.cuda:
            self.marks[-1].synchronize()
            return [a.elapsed_time(b) for a, b in zip(self.marks[:-1], self.marks[1:])]
        return [(b - a) * 1e3 for a, b in zip(self.marks[:-1], self.marks[1:])]


def batchdecode_sharded(decode_fn: Callable, syndromes: Optional[torch.Tensor], s: int, n: int, *,
                        root: int = 0, group=None, device: Optional[torch.device] = None,
                        comm_device: Optional[torch.device] = None, timing: Optional[Dict] = None):
    """Scatter -> local decode -> gather.

    decode_fn(syn [b][s] uint8 on `device`, out=(errors [b][n] uint8, converged [b] uint8, iters [b] int32))
    fills `out` (tensors on `device`) for its rows.
    `syndromes` ([B][s] uint8 on `device`) is read on the root only.  Returns
    (errors [B][n], converged [B], iters [B]) on the root and None elsewhere.

    `device`: where syndromes / results live (default: the current HIP device under RCCL, the host under
    gloo).  `comm_device`: where the exchanged buffers live when that differs -- ``torch.device("cpu")`` with a
    gloo group stages device shards through the host (the rehearsal of several ranks on one GPU).
    `timing`: a dict that receives ``scatter_ms``, ``decode_ms``, ``gather_ms`` of this rank for this call.
    """
    have_group = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_group else 0
    world = dist.get_world_size(group) if have_group else 1
    if device is None:
        device = syndromes.device if syndromes is not None else _default_device(group)
    device = torch.device(device)
    staged = comm_device is not None and torch.device(comm_device) != device
    cdev = torch.device(comm_device) if staged else device
    if rank == root:
        assert syndromes is not None and syndromes.dtype == torch.uint8 and syndromes.dim() == 2
        assert syndromes.shape[1] == s and syndromes.is_contiguous() and syndromes.device == device
    if world > 1:
        meta = torch.zeros(1, dtype=torch.int64, device=cdev)
        if rank == root:
            meta[0] = syndromes.shape[0]
        dist.broadcast(meta, src=root, group=group)
        B = int(meta.item())
    else:
        B = int(syndromes.shape[0])
    bounds = shard_bounds(B, world)
    lo, hi = bounds[rank]
    clock = _Clock(device) if timing is not None else None
    if clock:
        clock.mark()

    # ---- scatter the syndrome shards (the root keeps its own slice, no copy)
    if rank == root:
        mine = syndromes[lo:hi]
        src = syndromes.to(cdev) if staged else syndromes
        ops = [dist.P2POp(dist.isend, src[a:b], g, group) for g, (a, b) in enumerate(bounds)
               if g != root and b > a]
    else:
        mine_c = torch.empty((hi - lo, s), dtype=torch.uint8, device=cdev)
        ops = [dist.P2POp(dist.irecv, mine_c, root, group)] if hi > lo else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != root:
        mine = mine_c.to(device) if staged else mine_c
    if clock:
        clock.mark()

    # ---- local decode (the hot path; no communication inside).  The root decodes straight into its rows
    #      of the result arrays.
    if rank == root:
        errors = torch.empty((B, n), dtype=torch.uint8, device=device)
        converged = torch.empty(B, dtype=torch.uint8, device=device)
        iters = torch.empty(B, dtype=torch.int32, device=device)
        out = (errors[lo:hi], converged[lo:hi], iters[lo:hi])
    else:
        out = (torch.empty((hi - lo, n), dtype=torch.uint8, device=device),
               torch.empty(hi - lo, dtype=torch.uint8, device=device),
               torch.empty(hi - lo, dtype=torch.int32, device=device))
    if hi > lo:
        decode_fn(mine, out=out)
    if clock:
        clock.mark()

    # ---- gather errors / converged / iteration counts on the root
    if rank == root:
        ops, landing = [], []
        for g, (a, b) in enumerate(bounds):
            if g == root or b <= a:
                continue
            if staged:
                bufs = (torch.empty((b - a, n), dtype=torch.uint8, device=cdev),
                        torch.empty(b - a, dtype=torch.uint8, device=cdev),
                        torch.empty(b - a, dtype=torch.int32, device=cdev))
                landing.append((a, b, bufs))
            else:
                bufs = (errors[a:b], converged[a:b], iters[a:b])
            ops += [dist.P2POp(dist.irecv, t, g, group) for t in bufs]
    else:
        ops, landing = [], []
        if hi > lo:
            ops = [dist.P2POp(dist.isend, (t.to(cdev) if staged else t).contiguous(), root, group) for t in out]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for a, b, bufs in landing:
        errors[a:b].copy_(bufs[0])
        converged[a:b].copy_(bufs[1])
        iters[a:b].copy_(bufs[2])
    if clock:
        clock.mark()
        sc, de, ga = clock.spans_ms()
        timing.update(scatter_ms=sc, decode_ms=de, gather_ms=ga)
    if rank == root:
        return errors, converged, iters
    return None


def gpu_decode_fn(decoder) -> Callable:
    """decode_fn for batchdecode_sharded backed by a BeliefPropagationDecoder handle (HBM-resident I/O on torch's
    current stream).  The decode is asynchronous, but every rank must know that ITS shard is good before it takes
    part in the gather: a team of workgroups that was incomplete at launch or lost a member (other work on the GPU:
    ldpc_bp_last_status) leaves invalid outputs, and a rank that found out only at its next call would drop out of
    that step's exchange while the root waits for it.  So the rank waits for its decode, and on a fault decodes the
    shard once more -- the handle keeps teams off from then on, so the second pass cannot fail the same way."""
    from ._capi import LdpcError

    def fn(syn: torch.Tensor, out):
        err, conv, its = out
        syn = syn.contiguous()
        decoder.decode_batch_device(syn, err, conv, None, its)
        try:
            decoder.last_status()
        except LdpcError:
            decoder.decode_batch_device(syn, err, conv, None, its)
            decoder.last_status()

    return fn
