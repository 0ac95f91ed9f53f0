"""batchdecode! across the GPUs of one node: one process per GPU, contiguous batch shards.

The columns of `syndromes` are decoded independently (belief_propagation.jl:224-228),
so the path shards with NO collective inside the decode: the root hands every rank a
contiguous run of columns, each rank decodes its run with its own decoder handle, and
the hard decisions / flags travel back.  Exchange = grouped point-to-point
send/recv (``torch.distributed.batch_isend_irecv``; on the ``nccl`` backend that is an
RCCL ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI, each peer on its own
direct link to the root; on ``gloo`` the same code runs on CPU tensors for the tests).

When every rank already owns its syndromes (bench.py, production ingest from pinned host
memory per GPU) nothing here is needed: ranks simply call the decoder on their shard.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous shards: rank g gets columns [g*B/G, (g+1)*B/G)."""
    return [((batch * g) // world, (batch * (g + 1)) // world) for g in range(world)]


def batchdecode_sharded(decode_fn: Callable, syndromes: Optional[torch.Tensor], s: int, n: int, *,
                        root: int = 0, group=None, device: Optional[torch.device] = None):
    """Scatter -> local decode -> gather.

    decode_fn(syn [b][s] uint8 tensor on `device`) -> (errors [b][n] uint8, converged [b] uint8,
    iters [b] int32), all on `device`.
    `syndromes` ([B][s] uint8 on `device`) is read on the root only.  Returns
    (errors [B][n], converged [B], iters [B]) on the root and None elsewhere.
    """
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if device is None:
        device = syndromes.device if syndromes is not None else torch.device("cpu")
    meta = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == root:
        assert syndromes is not None and syndromes.dtype == torch.uint8 and syndromes.dim() == 2
        assert syndromes.shape[1] == s and syndromes.is_contiguous()
        meta[0] = syndromes.shape[0]
    dist.broadcast(meta, src=root, group=group)
    B = int(meta.item())
    bounds = shard_bounds(B, world)
    lo, hi = bounds[rank]

    # ---- scatter the syndrome shards (root keeps its own slice, no copy)
    if rank == root:
        mine = syndromes[lo:hi]
        ops = [dist.P2POp(dist.isend, syndromes[a:b], g, group) for g, (a, b) in enumerate(bounds)
               if g != root and b > a]
    else:
        mine = torch.empty((hi - lo, s), dtype=torch.uint8, device=device)
        ops = [dist.P2POp(dist.irecv, mine, root, group)] if hi > lo else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    # ---- local decode (the hot path; no communication inside)
    if hi > lo:
        err, conv, its = decode_fn(mine)
    else:
        err = torch.empty((0, n), dtype=torch.uint8, device=device)
        conv = torch.empty(0, dtype=torch.uint8, device=device)
        its = torch.empty(0, dtype=torch.int32, device=device)

    # ---- gather errors / converged / iteration counts on the root
    if rank == root:
        errors = torch.empty((B, n), dtype=torch.uint8, device=device)
        converged = torch.empty(B, dtype=torch.uint8, device=device)
        iters = torch.empty(B, dtype=torch.int32, device=device)
        errors[lo:hi] = err
        converged[lo:hi] = conv
        iters[lo:hi] = its
        ops = []
        for g, (a, b) in enumerate(bounds):
            if g == root or b <= a:
                continue
            ops += [dist.P2POp(dist.irecv, errors[a:b], g, group),
                    dist.P2POp(dist.irecv, converged[a:b], g, group),
                    dist.P2POp(dist.irecv, iters[a:b], g, group)]
    else:
        errors = converged = iters = None
        ops = []
        if hi > lo:
            ops = [dist.P2POp(dist.isend, err.contiguous(), root, group),
                   dist.P2POp(dist.isend, conv.contiguous(), root, group),
                   dist.P2POp(dist.isend, its.contiguous(), root, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank == root:
        return errors, converged, iters
    return None


def gpu_decode_fn(decoder) -> Callable:
    """decode_fn for batchdecode_sharded backed by a BeliefPropagationDecoder handle (HBM-resident I/O)."""

    def fn(syn: torch.Tensor):
        b = syn.shape[0]
        err = torch.empty((b, decoder.n), dtype=torch.uint8, device=syn.device)
        conv = torch.empty(b, dtype=torch.uint8, device=syn.device)
        its = torch.empty(b, dtype=torch.int32, device=syn.device)
        decoder.decode_batch_device(syn.contiguous(), err, conv, None, its)
        torch.cuda.current_stream(syn.device).synchronize()
        return err, conv, its

    return fn
