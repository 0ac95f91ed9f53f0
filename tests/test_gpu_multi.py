"""batchdecode! partitioned over several GPUs from ONE process through the C ABI (ldpc_bp_create_multi /
ldpc_bp_decode_batch_multi[_device]; csrc/ldpc_multi.hip) -- what a Julia host reaches with `ccall`
(belief_propagation.jl:220-231: one call, one caller-held matrix).  The GPU box has one GPU, so the N > 1 control
flow is rehearsed with logical devices that share GPU 0 (shards exchanged with hipMemcpyPeerAsync), and the RCCL
calls of a real multi-GPU run (ncclCommInitAll, grouped ncclSend / ncclRecv) with a one-rank communicator through
which the shard travels to itself.  Everything is checked against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(H, per, iters):
    from oracle import BPOracle

    return BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=iters)


def _close_llr(llr, ollr):
    fin = np.isfinite(ollr)
    return np.array_equal(llr[~fin], ollr[~fin]) and (not fin.any() or np.max(np.abs(llr[fin] - ollr[fin])) <= 1e-5)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_host_form_c2_against_the_oracle(ldpc, gpu, devices):
    """BASELINE config 2 ((3,6)-regular n = 1008, batch 4096) through the host form: every shard through its own
    logical device's pipeline on a host thread of its own; hard decisions, flags, iteration counts exact, LLRs 1e-5."""
    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(1008, 4096, 0.03, seed=31))
    dec = ldpc.BeliefPropagationDecoder(H, 0.03, 50, devices=devices)
    err, conv, llr, its = dec.decode_batch_host(syn, want_llr=True, want_iters=True)
    oerr, oconv, ollr, oits = _oracle(H, 0.03, 50).batchdecode(syn)
    assert np.array_equal(err, oerr) and np.array_equal(conv, oconv) and np.array_equal(its, oits)
    assert _close_llr(llr, ollr)
    mi = dec.multi_info()
    assert mi.ndev == len(devices) and list(mi.devices[:len(devices)]) == devices
    # the reference-interface functions take the same decoder: batchdecode! on s x B / n x B matrices
    errors = np.zeros((1008, 300), dtype=bool)
    _, success = ldpc.batchdecode_(dec, syn[:300].T, errors)
    assert np.array_equal(errors.T.astype(np.uint8), oerr[:300]) and np.array_equal(success, oconv[:300].astype(bool))
    dec.close()


@pytest.mark.parametrize("devices,exchange", [([0], 0), ([0, 0], 0), ([0], 2)])
def test_root_device_form_c3_team_kernel_against_the_oracle(ldpc, gpu, devices, exchange):
    """The n = 16384 code of configs 3 / 4 with the batch resident in the root's HBM: 1,280 syndromes = 20 tiles, so
    every shard takes the team kernel (two logical devices on one GPU: their team grids run one after the other).
    exchange 2 on one device: the shard is sent to itself through RCCL.  Equal to a single-device call bit for bit,
    and to the oracle on a sample."""
    import torch

    n = 16384
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    B = 1280
    syn_h = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, 0.05, seed=77))
    syn = torch.from_numpy(syn_h).cuda()
    single = ldpc.BeliefPropagationDecoder(H, 0.05, 14, device=0)
    e0 = torch.empty((B, n), dtype=torch.uint8, device="cuda"); c0 = torch.empty(B, dtype=torch.uint8, device="cuda")
    i0 = torch.empty(B, dtype=torch.int32, device="cuda")
    single.decode_batch_device(syn, e0, c0, None, i0)
    single.last_status()
    assert single.info().last_kernel == 4
    single.close()
    dec = ldpc.BeliefPropagationDecoder(H, 0.05, 14, devices=devices, exchange=exchange)
    for rep in range(2):
        e1 = torch.full((B, n), 9, dtype=torch.uint8, device="cuda"); c1 = torch.full((B,), 9, dtype=torch.uint8, device="cuda")
        i1 = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        dec.decode_batch_device(syn, e1, c1, None, i1)
        dec.last_status()
        torch.cuda.synchronize()
        assert torch.equal(e1, e0) and torch.equal(c1, c0) and torch.equal(i1, i0)
    mi = dec.multi_info()
    want = {(1, 0): ldpc._capi.EXCHANGE_NONE, (2, 0): ldpc._capi.EXCHANGE_COPY, (1, 2): ldpc._capi.EXCHANGE_RCCL}[(len(devices), exchange)]
    assert mi.exchange == want and mi.decode_ms_max > 0
    if len(devices) > 1:
        assert mi.scatter_bytes_per_peer == (B // 2) * H.shape[0] and mi.gather_bytes_per_peer == (B // 2) * (n + 1 + 4)
    k = 96                                   # oracle sample: the first and the last syndromes (both shards)
    idx = np.r_[0:k // 2, B - k // 2:B]
    oerr, oconv, _, oits = _oracle(H, 0.05, 14).batchdecode(syn_h[idx], want_llr=False)
    assert np.array_equal(e1.cpu().numpy()[idx], oerr) and np.array_equal(c1.cpu().numpy()[idx], oconv)
    assert np.array_equal(i1.cpu().numpy()[idx], oits)
    dec.close()


def test_root_device_form_with_llrs_and_ragged_shards(ldpc, gpu):
    """Three logical devices, a batch that does not divide (1000 = 333 + 333 + 334), LLRs gathered too; and a batch
    smaller than the device count (empty shards)."""
    import torch

    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    syn_h = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(1008, 1000, 0.04, seed=5))
    dec = ldpc.BeliefPropagationDecoder(H, 0.04, 30, devices=[0, 0, 0])
    orc = _oracle(H, 0.04, 30)
    for B in (1000, 2, 1):
        syn = torch.from_numpy(syn_h[:B]).cuda()
        e = torch.empty((B, 1008), dtype=torch.uint8, device="cuda"); c = torch.empty(B, dtype=torch.uint8, device="cuda")
        it = torch.empty(B, dtype=torch.int32, device="cuda"); llr = torch.empty((B, 1008), dtype=torch.float64, device="cuda")
        dec.decode_batch_device(syn, e, c, llr, it)
        dec.last_status()
        torch.cuda.synchronize()
        oerr, oconv, ollr, oits = orc.batchdecode(syn_h[:B])
        assert np.array_equal(e.cpu().numpy(), oerr) and np.array_equal(c.cpu().numpy(), oconv)
        assert np.array_equal(it.cpu().numpy(), oits) and _close_llr(llr.cpu().numpy(), ollr)
    dec.close()


def test_eight_way_control_flow_on_one_gpu(ldpc, gpu):
    """First-contact rehearsal for BASELINE config 4 (one caller-held matrix over the 8 GPUs of a node): EIGHT logical
    devices, all on GPU 0, through both forms of the one-process host -- shard_bounds for G = 8, seven peers' shard
    buffers and streams, the scatter to and the gather from every peer (hipMemcpyPeerAsync here: logical devices that
    share a GPU cannot form an RCCL clique; the RCCL calls themselves: the self-exchange test above).  The n = 16384
    code at a small batch, so that every shard is a few tiles of the team kernel; a ragged batch; LLRs and iteration
    counts gathered too.  Bit-equal to a single-device decode and to the oracle on a sample of every shard.  No N > 1
    number is taken from this."""
    import torch

    n = 16384
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    B = 8 * 160 + 5
    per, iters = 0.05, 12
    syn_h = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, per, seed=808))
    syn = torch.from_numpy(syn_h).cuda()
    single = ldpc.BeliefPropagationDecoder(H, per, iters, device=0)
    e0 = torch.empty((B, n), dtype=torch.uint8, device="cuda"); c0 = torch.empty(B, dtype=torch.uint8, device="cuda")
    i0 = torch.empty(B, dtype=torch.int32, device="cuda"); l0 = torch.empty((B, n), dtype=torch.float64, device="cuda")
    single.decode_batch_device(syn, e0, c0, l0, i0)
    single.last_status()
    single.close()
    dec = ldpc.BeliefPropagationDecoder(H, per, iters, devices=[0] * 8)
    mi = dec.multi_info()
    assert mi.ndev == 8 and mi.exchange == ldpc._capi.EXCHANGE_COPY
    # root-device form
    e1 = torch.full((B, n), 9, dtype=torch.uint8, device="cuda"); c1 = torch.full((B,), 9, dtype=torch.uint8, device="cuda")
    i1 = torch.full((B,), -1, dtype=torch.int32, device="cuda"); l1 = torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda")
    dec.decode_batch_device(syn, e1, c1, l1, i1)
    dec.last_status()
    torch.cuda.synchronize()
    assert torch.equal(e1, e0) and torch.equal(c1, c0) and torch.equal(i1, i0) and torch.equal(l1.view(torch.int64), l0.view(torch.int64))
    mi = dec.multi_info()
    lo7, hi7 = B * 7 // 8, B                                 # the last shard is the largest one of a ragged batch
    assert mi.scatter_bytes_per_peer == (hi7 - lo7) * H.shape[0] and mi.gather_bytes_per_peer == (hi7 - lo7) * (n + 1 + 4 + 8 * n)
    # host form: eight host threads, eight pipelines
    err, conv, llr, its = dec.decode_batch_host(syn_h, want_llr=True, want_iters=True)
    assert np.array_equal(err, e0.cpu().numpy()) and np.array_equal(conv, c0.cpu().numpy()) and np.array_equal(its, i0.cpu().numpy())
    assert np.array_equal(llr.view(np.int64), l0.cpu().numpy().view(np.int64))
    # the oracle on three syndromes of every shard
    idx = np.concatenate([np.arange(B * g // 8, B * g // 8 + 3) for g in range(8)])
    oerr, oconv, ollr, oits = _oracle(H, per, iters).batchdecode(syn_h[idx])
    assert np.array_equal(err[idx], oerr) and np.array_equal(conv[idx], oconv) and np.array_equal(its[idx], oits) and _close_llr(llr[idx], ollr)
    dec.close()
