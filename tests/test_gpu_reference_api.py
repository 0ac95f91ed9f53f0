"""The reference's own tests for the BP path, re-run against the MI355X decoder
through the host mirror of its API (``!`` spelled ``_``):

    test/test_bp_decoder.jl:1-52      exact recovery, batch LER, sequential LER
    src/decoders/belief_propagation.jl doctests (:76-82, :104-119, :204-218)
    test/test_bpots.jl:155-167        any AbstractVector is accepted (BitVector there)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _H(ldpc):
    return ldpc.parity_check_matrix(1000, 10, 9)   # test_bp_decoder.jl:7


def test_bp_decoder_single(ldpc, gpu):
    """test_bp_decoder.jl:6-16 -- guess == err."""
    rng = np.random.default_rng(1)
    H = _H(ldpc)
    per = 0.01
    err = rng.random(1000) < per
    syn = (H.astype(np.int64) @ err.astype(np.int64)) % 2      # Vector{Int}
    bpd = ldpc.BeliefPropagationDecoder(H, per, 100)
    guess, success = ldpc.decode_(bpd, syn)
    assert success is True
    assert guess is bpd.scratch.err and guess.dtype == np.float64   # alias of the scratch (:187)
    assert np.array_equal(guess, err.astype(np.float64))


def test_bp_decoder_batch(ldpc, gpu):
    """test_bp_decoder.jl:19-30 -- batchdecode! into zero(errors) (a BitMatrix), LER < 0.005."""
    rng = np.random.default_rng(2)
    H = _H(ldpc)
    per, num_trials = 0.01, 100
    errors = np.asfortranarray(rng.random((1000, num_trials)) < per)
    syndromes = (H.astype(np.int64) @ errors.astype(np.int64)) % 2   # Matrix{Int}, n x B
    bpd = ldpc.BeliefPropagationDecoder(H, per, 100)
    out = np.zeros_like(errors)
    guesses, successes = ldpc.batchdecode_(bpd, syndromes, out)
    assert guesses is out and successes.dtype == np.bool_ and successes.shape == (num_trials,)
    actual = [np.array_equal(guesses[:, i], errors[:, i]) for i in range(num_trials)]
    ler = (num_trials - sum(actual)) / num_trials
    assert ler < 0.005
    assert successes.all()


def test_ldpcdecoder_sequential(ldpc, gpu):
    """test_bp_decoder.jl:32-43 -- many decode! calls on one decoder (200 here), LER < 0.001."""
    rng = np.random.default_rng(3)
    H = _H(ldpc)
    dec = ldpc.BeliefPropagationDecoder(H, 0.01, 100)
    count = 0
    trials = 200
    for _ in range(trials):
        error = rng.random(1000) < 0.01
        syndrome = (H.astype(np.int64) @ error.astype(np.int64)) % 2
        guess, success = ldpc.decode_(dec, syndrome)
        count += np.array_equal(error, guess.astype(bool))
    assert 1 - count / trials < 0.001


def test_reset_and_doctests(ldpc, gpu):
    """belief_propagation.jl:76-82 (reset! callable, returns the decoder), :204-218 (10 samples)."""
    rng = np.random.default_rng(42)
    H = _H(ldpc)
    decoder = ldpc.BeliefPropagationDecoder(H, 0.01, 100)
    assert ldpc.reset_(decoder) is decoder
    errors = rng.random((1000, 10)) < 0.01
    syndromes = (H.astype(np.int64) @ errors.astype(np.int64)) % 2
    guesses, successes = ldpc.batchdecode_(decoder, syndromes, np.zeros_like(errors))
    assert guesses.shape == (1000, 10) and len(successes) == 10
    # the scratch is left holding the last column (the reference decodes columns in order)
    assert np.array_equal(decoder.scratch.err.astype(bool), guesses[:, -1])
    assert np.all(np.isfinite(decoder.scratch.log_probabs))


@pytest.mark.parametrize("kind", ["bool", "int64", "float64", "uint8", "view"])
def test_syndrome_element_types(ldpc, gpu, kind):
    """decode! takes any AbstractVector (Bool, Int, Float 0.0/1.0, views)."""
    rng = np.random.default_rng(4)
    H = ldpc.parity_check_matrix(504, 6, 3)
    dec = ldpc.BeliefPropagationDecoder(H, 0.02, 50)
    e = rng.random(504) < 0.02
    syn = (H.astype(np.int64) @ e.astype(np.int64)) % 2
    ref, ok_ref = ldpc.decode_(dec, syn)
    ref = ref.copy()
    if kind == "view":
        big = np.zeros((2, syn.size * 2), dtype=np.int64)
        big[1, ::2] = syn
        arg = big[1, ::2]
    else:
        arg = syn.astype(kind)
    got, ok = ldpc.decode_(dec, arg)
    assert ok == ok_ref and np.array_equal(got, ref)


def test_errors_eltype_and_success_vector(ldpc, gpu):
    """errors may be Bool or Int (test_bpots.jl:145 uses Matrix{Int}); a supplied success vector is filled."""
    rng = np.random.default_rng(5)
    H = ldpc.parity_check_matrix(504, 6, 3)
    dec = ldpc.BeliefPropagationDecoder(H, 0.02, 50)
    E = rng.random((504, 33)) < 0.02
    S = (H.astype(np.int64) @ E.astype(np.int64)) % 2
    out_i = np.zeros((504, 33), dtype=np.int64)
    succ = np.zeros(33, dtype=np.bool_)
    g, sc = ldpc.batchdecode_(dec, S, out_i, succ)
    assert g is out_i and sc is succ and succ.all()
    assert np.array_equal(out_i.astype(bool), E)
    out_f = np.zeros((504, 33), dtype=np.float64)
    ldpc.batchdecode_(dec, S.astype(bool), out_f)
    assert np.array_equal(out_f, out_i.astype(np.float64))


def test_batch_size_mismatch_is_an_assertion(ldpc, gpu):
    """@assert at belief_propagation.jl:221-222."""
    H = ldpc.parity_check_matrix(96, 6, 3)
    dec = ldpc.BeliefPropagationDecoder(H, 0.02, 5)
    with pytest.raises(AssertionError):
        ldpc.batchdecode_(dec, np.zeros((48, 4), dtype=np.int64), np.zeros((96, 5), dtype=bool))
    with pytest.raises(AssertionError):
        ldpc.batchdecode_(dec, np.zeros((48, 4), dtype=np.int64), np.zeros((96, 4), dtype=bool),
                          np.zeros(3, dtype=bool))


def test_empty_batch(ldpc, gpu):
    H = ldpc.parity_check_matrix(96, 6, 3)
    dec = ldpc.BeliefPropagationDecoder(H, 0.02, 5)
    g, s = ldpc.batchdecode_(dec, np.zeros((48, 0), dtype=np.int64), np.zeros((96, 0), dtype=bool))
    assert g.shape == (96, 0) and s.shape == (0,)


def test_invalid_csc_is_rejected(ldpc, gpu):
    import ctypes

    L = ldpc._capi.lib()
    colptr = np.array([0, 2, 2], dtype=np.int64)
    rowval = np.array([1, 0], dtype=np.int64)   # not ascending inside the column
    h = ctypes.c_void_p()
    st = L.ldpc_bp_create(2, 2, 2, colptr.ctypes.data, rowval.ctypes.data, 0.1, 5, None, ctypes.byref(h))
    assert st == 1 and b"ascending" in L.ldpc_last_error()


def test_large_host_batch_goes_through_the_chunked_pipeline(ldpc, gpu):
    """Host arrays far larger than one pipeline chunk (pinned staging, 3 slots, ragged tail):
    results must equal the HBM-resident entry row for row, and the oracle on a subset."""
    import torch

    from oracle import BPOracle

    H = ldpc.codes.parity_check_csc(504, 6, 3)
    B = 150001
    E = ldpc.codes.random_errors(504, B, 0.02, seed=8)
    syn = ldpc.codes.syndromes_of(H, E)
    dec = ldpc.BeliefPropagationDecoder(H, 0.02, 40)
    for want_llr in (False, True):
        err, conv, llr, its = dec.decode_batch_host(syn, want_llr=want_llr, want_iters=True)
        d_syn = torch.from_numpy(syn).cuda()
        d_err = torch.empty((B, 504), dtype=torch.uint8, device="cuda")
        d_conv = torch.empty(B, dtype=torch.uint8, device="cuda")
        d_it = torch.empty(B, dtype=torch.int32, device="cuda")
        d_llr = torch.empty((B, 504), dtype=torch.float64, device="cuda") if want_llr else None
        dec.decode_batch_device(d_syn, d_err, d_conv, d_llr, d_it)
        torch.cuda.synchronize()
        assert np.array_equal(err, d_err.cpu().numpy())
        assert np.array_equal(conv, d_conv.cpu().numpy()) and np.array_equal(its, d_it.cpu().numpy())
        if want_llr:
            assert np.array_equal(llr, d_llr.cpu().numpy(), equal_nan=True)
    idx = np.random.default_rng(0).choice(B, 3000, replace=False)
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.02, max_iters=40)
    oerr, oconv, _, oits = oc.batchdecode(syn[idx], want_llr=False)
    assert np.array_equal(err[idx], oerr) and np.array_equal(conv[idx], oconv) and np.array_equal(its[idx], oits)


def test_decoder_lifecycle_does_not_leak_device_memory(ldpc, gpu):
    """Create / decode / destroy many decoders of every kind (LDS, node-parallel, tile kernels, BP-OTS,
    BP+OSD; latency path and staged path): device memory returns to where it started, and handles of
    different kinds can be alive at the same time."""
    import torch

    H1 = ldpc.codes.parity_check_csc(1008, 6, 3)
    H2 = ldpc.codes.parity_check_csc(4096, 8, 4)
    S1 = ldpc.codes.syndromes_of(H1, ldpc.codes.random_errors(1008, 300, 0.01, seed=1))
    S2 = ldpc.codes.syndromes_of(H2, ldpc.codes.random_errors(4096, 300, 0.02, seed=2))

    def round_trip():
        a = ldpc.BeliefPropagationDecoder(H1, 0.01, 50)
        b = ldpc.BeliefPropagationDecoder(H2, 0.02, 50)                     # node-parallel for small batches
        c = ldpc.BeliefPropagationDecoder(H2, 0.02, 50, kernel_variant=1)    # tile kernel
        o = ldpc.BPOTSDecoder(H1, 0.01, 30)
        q = ldpc.BeliefPropagationOSDDecoder(H1, 0.01, 20, osd_order=2)
        r1 = a.decode_batch_host(S1[:1])            # latency path
        r2 = a.decode_batch_host(S1, want_llr=True)  # staged path
        r3 = b.decode_batch_host(S2[:3])
        r4 = c.decode_batch_host(S2[:130], want_llr=True)
        o.decode_batch_host(S1[:2])
        q.decode_(S1[0])
        assert np.array_equal(r1[0][0], r2[0][0]) and np.array_equal(r3[0], r4[0][:3])
        for d in (a, b, c, o):
            d.close()
        del q

    round_trip()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        round_trip()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), f"device memory shrank by {(free0 - free1) >> 20} MiB over 25 decoder life cycles"


def test_timing_history_reaches_sixteen_calls_back(ldpc, gpu):
    """ldpc_bp_call_timing: the last 16 batch calls keep their HIP-event times and iteration sums (bench.py
    reads them after its timed region); one further back is an argument error, not stale data."""
    import torch

    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    dec = ldpc.BeliefPropagationDecoder(H, 0.01, 50)
    B = 700
    sums = []
    for k in range(20):
        syn = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(1008, B, 0.01, seed=k))).cuda()
        err = torch.empty((B, 1008), dtype=torch.uint8, device="cuda")
        conv = torch.empty(B, dtype=torch.uint8, device="cuda")
        its = torch.empty(B, dtype=torch.int32, device="cuda")
        dec.decode_batch_device(syn, err, conv, None, its)
        torch.cuda.synchronize()
        sums.append(int(its.sum().item()))
    for back in range(16):
        sweep_ms, total_ms, sum_iters = dec.last_timing(back)
        assert sum_iters == sums[19 - back] and 0 < sweep_ms <= total_ms
    with pytest.raises(ldpc.LdpcError):
        dec.last_timing(16)


def test_two_handles_from_two_threads(ldpc, gpu):
    """"Different handles may be used from different threads" (include/ldpc_mi355x.h): two decoders, each
    hammered from its own host thread with single decode! calls (latency path), small and medium batches
    (staged path; LDS, node-parallel and team kernels, the latter launched cooperatively) -- every result
    must equal the one the same handle gives when it runs alone."""
    import threading

    H1 = ldpc.codes.parity_check_csc(1008, 6, 3)
    H2 = ldpc.codes.parity_check_csc(16384, 8, 4)
    S1 = ldpc.codes.syndromes_of(H1, ldpc.codes.random_errors(1008, 900, 0.02, seed=5))
    S2 = ldpc.codes.syndromes_of(H2, ldpc.codes.random_errors(16384, 300, 0.02, seed=6))
    d1 = ldpc.BeliefPropagationDecoder(H1, 0.02, 50)
    d2 = ldpc.BeliefPropagationDecoder(H2, 0.02, 50)
    plan1 = [slice(k, k + 1) for k in range(40)] + [slice(0, 900), slice(100, 164)]
    plan2 = [slice(k, k + 1) for k in range(6)] + [slice(0, 300), slice(0, 64), slice(10, 140)]
    ref1 = [d1.decode_batch_host(S1[sl], want_iters=True) for sl in plan1]
    ref2 = [d2.decode_batch_host(S2[sl], want_iters=True) for sl in plan2]
    errors = []

    def worker(dec, S, plan, ref):
        try:
            for rep in range(3):
                for sl, want in zip(plan, ref):
                    got = dec.decode_batch_host(S[sl], want_iters=True)
                    if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[3], want[3])):
                        errors.append((S.shape, sl))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(d1, S1, plan1, ref1)), threading.Thread(target=worker, args=(d2, S2, plan2, ref2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_two_c3_size_handles_from_two_threads(ldpc, gpu):
    """Two decoders of the C3 code (n = 16384) on the tile kernel (kernel_variant 1: one workgroup per tile, 768 tiles
    in flight; the default path for this code -- persistent teams -- needs 8 message slots only), each with a
    full-size batch of its own -- 65,536 syndromes, so each
    allocates the 24.75 GiB message workspace, the packed hand-off levels, and may run its placement search (up to
    five 1 GiB-chunk groups held at once, never beyond half of the free HBM) -- created and driven concurrently from
    two host threads on two streams.  Neither may starve the other of memory, and both must produce what a lone
    decoder produces."""
    import threading

    import torch

    n, B = 16384, 65536
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    csr = H.tocsr()
    dev = torch.device("cuda:0")
    cols = torch.from_numpy(csr.indices.astype(np.int64)).to(dev)

    def make(seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        syn = torch.empty((B, H.shape[0]), dtype=torch.uint8, device=dev)
        for b0 in range(0, B, 4096):
            e = (torch.rand((4096, n), generator=g, device=dev) < 0.02).to(torch.uint8)
            syn[b0:b0 + 4096] = e[:, cols].view(4096, H.shape[0], 8).sum(dim=2, dtype=torch.int32).remainder(2).to(torch.uint8)
        return syn

    syns = [make(101), make(202)]
    ldpc._capi.check(ldpc._capi.lib().ldpc_trim_memory())
    torch.cuda.synchronize()
    free_start = torch.cuda.mem_get_info()[0]
    lone = []
    for syn in syns:                                   # the reference: one decoder at a time
        dec = ldpc.BeliefPropagationDecoder(H, 0.02, 50, kernel_variant=1)
        e = torch.empty((B, n), dtype=torch.uint8, device=dev)
        c = torch.empty(B, dtype=torch.uint8, device=dev)
        i = torch.empty(B, dtype=torch.int32, device=dev)
        dec.decode_batch_device(syn, e, c, None, i)
        torch.cuda.synchronize()
        lone.append((c.clone(), i.clone(), e.sum(dim=1, dtype=torch.int32)))
        del e
        dec.close()
    torch.cuda.synchronize()
    errors, results = [], [None, None]

    def worker(k):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            dec = ldpc.BeliefPropagationDecoder(H, 0.02, 50, kernel_variant=1)
            e = torch.empty((B, n), dtype=torch.uint8, device=dev)
            c = torch.empty(B, dtype=torch.uint8, device=dev)
            i = torch.empty(B, dtype=torch.int32, device=dev)
            st.wait_stream(torch.cuda.current_stream())
            for _ in range(2):
                dec.decode_batch_device(syns[k], e, c, None, i, stream=st.cuda_stream)
            dec.last_status()
            results[k] = (c, i, e.sum(dim=1, dtype=torch.int32))
            assert dec.info().workspace_bytes > 24 * (1 << 30)
            dec.close()
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for k in range(2):
        assert all(torch.equal(a, b) for a, b in zip(results[k], lone[k])), k
    del results, lone
    torch.cuda.empty_cache()
    # the chunk groups of closed decoders wait in the library's pool (up to 64 GiB) for the next decoder of that size;
    # ldpc_trim_memory gives them back
    assert torch.cuda.mem_get_info()[0] < free_start - (20 << 30)
    ldpc._capi.check(ldpc._capi.lib().ldpc_trim_memory())
    assert torch.cuda.mem_get_info()[0] >= free_start - (2 << 30)     # every group, level and candidate was given back


def test_a_host_side_wait_that_expires_names_itself(gpu):
    """csrc/host_wait.hpp: every wait of the host for the device is bounded (ldpc_set_wait_limit_ms).  With the limit
    at 20 ms a decode that needs hundreds of milliseconds on the device (the n = 16384 code, 50 iterations, 4,096
    syndromes through the synchronous host entry) must come back with LDPC_ERR_HIP NAMING the wait instead of blocking;
    the device then counts as stalled for the process: the next call fails at once with the same message, and closing
    the decoder returns (what the device may still use is leaked, not freed under it).  In a process of its own, since
    the mark is for the life of the process; that process must still exit by itself."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import sys, time
        import numpy as np
        import ldpcdecoders_jl_amd as ldpc
        L = ldpc._capi.lib()
        H = ldpc.codes.parity_check_csc(16384, 8, 4)
        syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(16384, 4096, 0.10, seed=1))
        dec = ldpc.BeliefPropagationDecoder(H, 0.10, 50)
        dec.decode_batch_host(syn)                            # (first call: workspace, tables, code objects -- under the default limit;
        assert L.ldpc_set_wait_limit_ms(20) == 0              #  a first call spends tens of ms loading kernels while the device already computes)
        t0 = time.time()
        try:
            dec.decode_batch_host(syn)
            print("NOT BOUNDED")
            sys.exit(3)
        except ldpc.LdpcError as e:
            msg = e.message
            assert e.status == 3 and "did not get there within" in msg and "ldpc_bp_decode_batch" in msg, msg
        assert time.time() - t0 < 5.0
        try:
            dec.decode_batch_host(syn[:64])
            sys.exit(4)
        except ldpc.LdpcError as e:
            assert e.message == msg, (e.message, msg)         # stalled: the same message, at once
        t0 = time.time()
        dec.close()
        assert time.time() - t0 < 5.0
        print("BOUNDED", msg[:60])
        """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         cwd=__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    assert out.returncode == 0 and "BOUNDED" in out.stdout and "NOT BOUNDED" not in out.stdout, (out.returncode, out.stdout[-500:], out.stderr[-1500:])
