"""The HIP decoder through ldpcdecoders.jl_amd/sharding.py (BASELINE config 4's path): world size 1 on the
RCCL backend (the degenerate case of the same code), and two ranks sharing this box's one GPU with the
exchange over gloo and host staging (bench.py --rehearse-on-one-gpu) -- the root-held batch is scattered,
decoded by both ranks' HIP decoders and gathered; bench.py --verify has the root decode every shard by
itself and compare."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_world_size_one_on_the_rccl_backend_equals_a_direct_call(ldpc, gpu):
    import torch
    import torch.distributed as dist

    from ldpcdecoders_jl_amd import sharding

    n = 16384
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    B = 1500
    syn = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, 0.03, seed=21))).cuda()
    dec = ldpc.BeliefPropagationDecoder(H, 0.03, 50)
    err = torch.empty((B, n), dtype=torch.uint8, device="cuda")
    conv = torch.empty(B, dtype=torch.uint8, device="cuda")
    its = torch.empty(B, dtype=torch.int32, device="cuda")
    dec.decode_batch_device(syn, err, conv, None, its)
    torch.cuda.synchronize()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        tm = {}
        e2, c2, i2 = sharding.batchdecode_sharded(sharding.gpu_decode_fn(dec), syn, H.shape[0], n, timing=tm)
        torch.cuda.synchronize()
        assert e2.device.type == "cuda"           # default device under RCCL = the current HIP device
        assert torch.equal(e2, err) and torch.equal(c2, conv) and torch.equal(i2, its)
        assert tm["decode_ms"] > 0 and tm["scatter_ms"] >= 0 and tm["gather_ms"] >= 0
        # a non-root default (no syndromes to take the device from) must land on the GPU too
        assert sharding._default_device(None).type == "cuda"
    finally:
        dist.destroy_process_group()
    # and without any process group
    e3, c3, i3 = sharding.batchdecode_sharded(sharding.gpu_decode_fn(dec), syn, H.shape[0], n)
    torch.cuda.synchronize()
    assert torch.equal(e3, err) and torch.equal(c3, conv) and torch.equal(i3, its)
    dec.close()


@pytest.mark.gpu
def test_two_ranks_scatter_decode_gather_with_the_hip_decoder(gpu):
    """bench.py --gpus 2 in its default (scatter) mode, both ranks on cuda:0: rank 0 holds the 2 x 3000 batch, the
    shards travel (gloo, staged through the host), both HIP decoders decode, the root gathers; --verify compares
    every gathered shard with a local decode on the root."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu",
           "--workload", "c3_realistic", "--batch", "3000", "--steps", "1", "--warmup", "1", "--verify", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 6000 and d["config"]["mode"] == "scatter"
    x = d["exchange"]
    assert x["sharded_matches_local"] is True
    assert x["scatter_bytes_per_peer"] == 3000 * 8192 and x["scatter_ms"] > 0 and x["gather_ms"] > 0 and x["decode_ms_max_rank"] > 0
    assert abs(d["value"] - 6000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_cabi_mode_two_logical_devices_one_process(gpu):
    """bench.py --gpus 2 --mode cabi --rehearse-on-one-gpu: ONE process, ldpc_bp_create_multi with two logical devices on
    GPU 0 -- the root-held 2 x 3000 batch is scattered, decoded on both handles' streams and gathered through the C ABI;
    the same `exchange` fields as the torch.distributed mode; --verify compares with a single-device decoder."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "cabi", "--rehearse-on-one-gpu",
           "--workload", "c3_realistic", "--batch", "3000", "--steps", "1", "--warmup", "1", "--verify", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 6000 and d["config"]["mode"] == "cabi"
    x = d["exchange"]
    assert x["sharded_matches_local"] is True
    assert x["scatter_bytes_per_peer"] == 3000 * 8192 and x["gather_bytes_per_peer"] == 3000 * (16384 + 1 + 4)
    assert x["scatter_ms"] >= 0 and x["gather_ms"] > 0 and x["decode_ms_max_rank"] > 0
    assert abs(d["value"] - 6000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
