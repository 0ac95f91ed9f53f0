"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the scatter -> decode -> gather
of ldpcdecoders.jl_amd/sharding.py, with the oracle standing in for the per-rank HIP decoder
(the HIP decoder itself goes through the same function in tests/test_gpu_sharding.py)."""
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


def _oracle_decode_fn(oc):
    def decode_fn(syn, out):
        err, conv, _, its = oc.batchdecode(syn.numpy(), want_llr=False)
        out[0].copy_(torch.from_numpy(err))
        out[1].copy_(torch.from_numpy(conv))
        out[2].copy_(torch.from_numpy(its))

    return decode_fn


def _worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H = ldpc.codes.parity_check_csc(504, 6, 3)
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.03, max_iters=30)
        syn = None
        if rank == 0:
            E = ldpc.codes.random_errors(504, B, 0.03, seed=B)
            syn = torch.from_numpy(ldpc.codes.syndromes_of(H, E))
        timing = {}
        res = sharding.batchdecode_sharded(_oracle_decode_fn(oc), syn, 252, 504, root=0, timing=timing)
        assert set(timing) == {"scatter_ms", "decode_ms", "gather_ms"} and all(v >= 0 for v in timing.values())
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


# (8, ...): BASELINE config 4's world size -- seven peers' shard buffers, 7 grouped sends in the scatter and 3 x 7 grouped
# receives in the gather on the root; a batch smaller than the world (empty shards) too
@pytest.mark.parametrize("world,B", [(2, 37), (2, 1), (3, 100), (1, 9), (8, 203), (8, 5)])
def test_scatter_decode_gather(tmp_path, world, B):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").read_text() == "1"


def test_no_process_group_is_world_size_one():
    """N = 1 is the degenerate case of the same code: without torch.distributed the root's
    shard is the whole batch and is decoded straight into the result arrays."""
    assert not dist.is_initialized()
    H = ldpc.codes.parity_check_csc(504, 6, 3)
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.03, max_iters=30)
    syn = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(504, 21, 0.03, seed=3)))
    err, conv, its = sharding.batchdecode_sharded(_oracle_decode_fn(oc), syn, 252, 504)
    ref = oc.batchdecode(syn.numpy(), want_llr=False)
    assert np.array_equal(err.numpy(), ref[0]) and np.array_equal(conv.numpy(), ref[1]) and np.array_equal(its.numpy(), ref[3])


def test_shard_bounds_cover_the_batch():
    for B in (0, 1, 7, 64, 65536, 524288):
        for G in (1, 2, 3, 8):
            b = sharding.shard_bounds(B, G)
            assert b[0][0] == 0 and b[-1][1] == B and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
