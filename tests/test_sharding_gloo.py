"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the scatter -> decode -> gather
of ldpcdecoders.jl_amd/sharding.py, with the oracle standing in for the per-rank HIP decoder."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ldpcdecoders_jl_amd as ldpc
from ldpcdecoders_jl_amd import sharding
from oracle import BPOracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H = ldpc.codes.parity_check_csc(504, 6, 3)
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.03, max_iters=30)

        def decode_fn(syn):
            err, conv, _, its = oc.batchdecode(syn.numpy(), want_llr=False)
            return torch.from_numpy(err), torch.from_numpy(conv), torch.from_numpy(its)

        syn = None
        if rank == 0:
            E = ldpc.codes.random_errors(504, B, 0.03, seed=B)
            syn = torch.from_numpy(ldpc.codes.syndromes_of(H, E))
        res = sharding.batchdecode_sharded(decode_fn, syn, 252, 504, root=0)
        if rank == 0:
            err, conv, its = res
            ref = oc.batchdecode(syn.numpy(), want_llr=False)
            ok = (np.array_equal(err.numpy(), ref[0]) and np.array_equal(conv.numpy(), ref[1])
                  and np.array_equal(its.numpy(), ref[3]))
            open(os.path.join(out_dir, "ok"), "w").write("1" if ok else "0")
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 37), (2, 1), (3, 100)])
def test_scatter_decode_gather(tmp_path, world, B):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").read_text() == "1"
