"""The reference's own acceptance tests for the BP path, run on the CPU oracle.

These are the only behavioural pins the reference holds for this path:
test/test_bp_decoder.jl:46-51 -- 1 + 100 + 1000 random trials on the (9,10)-regular
n=1000 code, per=0.01, 100 iterations, all must be recovered exactly; and
test/test_oldtests.jl:13-16 for the generator."""
import numpy as np

import ldpcdecoders_jl_amd as ldpc
from oracle import BPOracle


def _oracle(H, per, iters):
    return BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=iters)


def test_bp_decoder_single_batch_and_sequential():
    H = ldpc.codes.parity_check_csc(1000, 10, 9)
    oc = _oracle(H, 0.01, 100)
    rng = np.random.default_rng(2024)
    # test_bp_decoder(): one trial, guess == err                       (test_bp_decoder.jl:6-16,46)
    e = (rng.random(1000) < 0.01).astype(np.uint8)
    guess, ok = oc.decode(ldpc.codes.syndromes_of(H, e[None, :])[0])
    assert ok and np.array_equal(guess.astype(np.uint8), e)
    # test_bp_decoder_batch(): 100 columns, LER < 0.005 -> zero failures (:19-30,49)
    E = (rng.random((100, 1000)) < 0.01).astype(np.uint8)
    err, conv, _, _ = oc.batchdecode(ldpc.codes.syndromes_of(H, E), want_llr=False)
    assert (100 - int(np.all(err == E, axis=1).sum())) / 100 < 0.005
    assert conv.all()
    # test_ldpcdecoder(): 1000 sequential decodes on one decoder, LER < 0.001 (:32-43,51)
    E = (rng.random((1000, 1000)) < 0.01).astype(np.uint8)
    S = ldpc.codes.syndromes_of(H, E)
    count = 0
    for b in range(1000):
        g, _ = oc.decode(S[b])
        count += np.array_equal(g.astype(np.uint8), E[b])
    assert 1 - count / 1000 < 0.001


def test_dense_mode_is_value_identical_to_edge_list_mode():
    """The reference-faithful dense storage (cpu_baseline) and the edge list are the same arithmetic."""
    H = ldpc.codes.parity_check_csc(504, 6, 3)
    E = ldpc.codes.random_errors(504, 40, 0.05, seed=9)
    S = ldpc.codes.syndromes_of(H, E)
    a = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.05, max_iters=30, dense=False).batchdecode(S)
    b = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.05, max_iters=30, dense=True).batchdecode(S)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


def test_parity_check_matrix_generation():
    """test/test_oldtests.jl:1-17: all row sums = wr, all column sums = wc."""
    H = ldpc.parity_check_matrix(1000, 10, 9)
    assert H.shape == (900, 1000) and H.dtype == np.bool_
    assert np.all(H.sum(axis=1) == 10) and np.all(H.sum(axis=0) == 9)
    # structure of parity_generator.jl:32-36: block 0 = consecutive runs of wr ones
    assert all(H[i, i * 10:(i + 1) * 10].all() for i in range(100))
    # seeded: reproducible, and a different seed gives a different matrix
    assert np.array_equal(H, ldpc.parity_check_matrix(1000, 10, 9))
    assert not np.array_equal(H, ldpc.parity_check_matrix(1000, 10, 9, seed=1))
