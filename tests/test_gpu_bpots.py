"""BP-OTS on the MI355X (ldpc_bpots_*, LDS-resident kernel) against the CPU oracle: estimates,
convergence flags and iteration counts must be identical (both sides use the portable tanh/atanh),
plus the reference's own assertions from test/test_bpots.jl through the host mirror."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import BPOTSOracle

pytestmark = pytest.mark.gpu


def cycle_matrix(n):                       # test_bpots.jl:13-24
    H = np.zeros((n, n), dtype=np.uint8)
    for j in range(n):
        H[j, (j + 1) % n] = 1
        H[j, j] = 1
    return H


def toric_x(d):
    H = np.zeros((d * d, 2 * d * d), dtype=np.uint8)
    for r in range(d):
        for c in range(d):
            v = r * d + c
            H[v, r * d + c] = 1
            H[v, r * d + (c - 1) % d] = 1
            H[v, d * d + r * d + c] = 1
            H[v, d * d + ((r - 1) % d) * d + c] = 1
    return H


def assert_same_as_oracle(ldpc, H, per, iters, T, C, syn, kernel=None):
    M = sp.csc_matrix(H)
    M.sort_indices()
    oerr, oconv, oits = BPOTSOracle((M.indptr, M.indices), M.shape, per, iters, T, C).batchdecode(syn)
    dec = ldpc.BPOTSDecoder(M, per, iters, T=T, C=C)
    assert kernel is None or dec.kernel == kernel
    err, conv, its = dec.decode_batch_host(syn)
    bad = np.nonzero((err != oerr).any(axis=1) | (conv != oconv) | (its != oits))[0]
    assert bad.size == 0, f"{bad.size} of {syn.shape[0]} syndromes differ from the oracle, first {bad[:5]}"
    dec.close()
    return err, conv, its


@pytest.mark.parametrize("n", [4, 8, 16])
@pytest.mark.parametrize("T,C", [(3, 1.0), (5, 2.0), (9, 3.0)])
def test_trapping_set_resistance(ldpc, gpu, n, T, C):
    """test_bpots.jl:55-84."""
    H = cycle_matrix(n)
    e = np.zeros(n, dtype=np.uint8)
    e[:2] = 1
    syn = (H.astype(int) @ e % 2).astype(np.uint8)
    bpots = ldpc.BPOTSDecoder(H, 0.01, 100, T=T, C=C)
    result, converged = bpots.decode_(syn.astype(bool))
    assert np.array_equal(H.astype(int) @ result % 2, syn)
    rng = np.random.default_rng(n * 10 + T)
    S = (rng.integers(0, 2, (70, n)).astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
    assert_same_as_oracle(ldpc, H, 0.01, 100, T, C, S)


def test_parameter_sensitivity_and_batch(ldpc, gpu):
    """test_bpots.jl:87-113, :139-153: random syndromes of the 4- and 8-cycle must be reproduced."""
    rng = np.random.default_rng(3)
    H = cycle_matrix(4)
    for T, C in [(3, 3.0), (5, 3.0), (9, 3.0), (15, 3.0), (9, 1.0), (9, 2.0), (9, 5.0), (9, 10.0)]:
        syn = (rng.integers(0, 2, 4) @ H.T.astype(int) % 2).astype(np.uint8)
        result, converged = ldpc.BPOTSDecoder(H, 0.01, 100, T=T, C=C).decode_(syn)
        assert np.array_equal(H.astype(int) @ result % 2, syn)
    H = cycle_matrix(8)
    decoder = ldpc.BPOTSDecoder(H, 0.01, 100, T=9, C=3.0)
    syndromes = (H.astype(int) @ rng.integers(0, 2, (8, 5)) % 2)
    errors_buf = np.zeros((8, 5), dtype=np.int64)
    guesses, successes = ldpc.batchdecode_(decoder, syndromes, errors_buf)
    for i in range(5):
        assert np.array_equal(H.astype(int) @ guesses[:, i] % 2, syndromes[:, i])


def test_toric_code(ldpc, gpu):
    """test_bpots.jl:116-137 and equality with the oracle on every syndrome."""
    H = toric_x(3)
    rng = np.random.default_rng(7)
    for noise in (0.01, 0.05, 0.1):
        E = rng.integers(0, 2, (300, 18)).astype(np.uint8)
        syn = (E.astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
        err, conv, _ = assert_same_as_oracle(ldpc, H, noise, 50, 9, 3.0, syn)
        ok = np.all((err.astype(int) @ H.T.astype(int) % 2) == syn, axis=1)
        assert ok.mean() >= 0.85


@pytest.mark.parametrize("B", [1, 63, 257])
def test_ldpc_code_hard_syndromes(ldpc, gpu, B):
    """(3,6)-regular n=504 above threshold: many syndromes run into the bias step (iter % T == 0)."""
    H = ldpc.codes.parity_check_csc(504, 6, 3)
    E = ldpc.codes.random_errors(504, B, 0.06, seed=B)
    syn = ldpc.codes.syndromes_of(H, E)
    err, conv, its = assert_same_as_oracle(ldpc, H, 0.06, 60, 9, 2.0, syn)
    if B > 100:
        assert (its > 9).any()


def test_bb72_and_irregular_graphs(ldpc, gpu):
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    E = ldpc.codes.random_errors(72, 3000, 0.05, seed=2)
    assert_same_as_oracle(ldpc, HX, 0.05, 40, 5, 2.0, ldpc.codes.syndromes_of(HX, E))
    rng = np.random.default_rng(9)
    for trial in range(4):
        H = (rng.random((20, 40)) < 0.12).astype(np.uint8)
        H[0, :] = 0
        H[:, 1] = 0
        syn = rng.integers(0, 2, (130, 20)).astype(np.uint8)
        syn[3, 2] = 2
        assert_same_as_oracle(ldpc, H, 0.03, 30, 4, 1.5, syn)


def test_max_iters_zero(ldpc, gpu):
    H = cycle_matrix(8)
    err, conv, its = ldpc.BPOTSDecoder(H, 0.01, 0).decode_batch_host(np.zeros((3, 8), dtype=np.uint8))
    assert not err.any() and not conv.any() and not its.any()


def test_no_size_or_degree_limit(ldpc, gpu):
    """The reference's BPOTSDecoder takes any H (bpots_decoder.jl:225-340); round 2 returned LDPC_ERR_UNSUPPORTED
    beyond n ~ 30,000 and for check degree > 32 / bit degree > 16.  Those graphs now take bpots_big_kernel (everything
    of a syndrome in a global slot, nodes of any degree; ldpc_bpots_kernel = 5): a (4,8)-regular n = 65536 code, a
    graph with a check of degree 40 and a bit of degree 20, and one with a few heavy nodes among light ones -- estimates,
    flags and iteration counts equal the oracle's."""
    huge = ldpc.codes.parity_check_csc(65536, 8, 4)
    syn = ldpc.codes.syndromes_of(huge, ldpc.codes.random_errors(65536, 6, 0.02, seed=4))
    err, conv, its = assert_same_as_oracle(ldpc, huge, 0.02, 12, 9, 2.0, syn, kernel=5)
    assert conv.all()
    rng = np.random.default_rng(21)
    wide = np.zeros((24, 60), dtype=np.uint8)
    wide[0, :40] = 1                                               # check degree 40 > 32
    wide[:20, 59] = 1                                              # bit degree 20 > 16
    wide |= (rng.random((24, 60)) < 0.08).astype(np.uint8)
    S = rng.integers(0, 2, (200, 24)).astype(np.uint8)
    S[5, 3] = 3
    assert_same_as_oracle(ldpc, wide, 0.04, 40, 5, 2.0, S, kernel=5)
    E = (rng.random((150, 60)) < 0.05).astype(np.uint8)
    assert_same_as_oracle(ldpc, wide, 0.05, 30, 9, 1.5, (E.astype(int) @ wide.T.astype(int) % 2).astype(np.uint8), kernel=5)


def test_unlimited_kernel_equals_the_other_two_on_small_graphs(ldpc, gpu, monkeypatch):
    """LDPC_BPOTS_FORCE_NODE=2 sends everything through bpots_big_kernel: the graphs of the reference's BP-OTS tests and
    the n = 8190 code against the oracle (and thereby against the LDS-resident and the node-parallel kernel)."""
    monkeypatch.setenv("LDPC_BPOTS_FORCE_NODE", "2")
    rng = np.random.default_rng(13)
    for n in (4, 8, 16):
        H = cycle_matrix(n)
        S = (rng.integers(0, 2, (90, n)).astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
        assert_same_as_oracle(ldpc, H, 0.01, 100, 3, 1.0, S, kernel=5)
    H = toric_x(3)
    E = rng.integers(0, 2, (300, 18)).astype(np.uint8)
    assert_same_as_oracle(ldpc, H, 0.05, 50, 9, 3.0, (E.astype(int) @ H.T.astype(int) % 2).astype(np.uint8), kernel=5)
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    assert_same_as_oracle(ldpc, HX, 0.05, 40, 5, 2.0, ldpc.codes.syndromes_of(HX, ldpc.codes.random_errors(72, 500, 0.05, seed=2)), kernel=5)
    H = ldpc.codes.parity_check_csc(8190, 6, 3)
    syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(8190, 40, 0.07, seed=8))
    assert_same_as_oracle(ldpc, H, 0.07, 30, 9, 2.0, syn, kernel=5)


def test_gpu_against_the_independent_libm_oracle_stays_inside_the_band(ldpc, gpu):
    """GPU == oracle above is by construction on tanh / atanh (kernel and oracle share portable_math.h).  The INDEPENDENT
    witness is oracle/libbpots_oracle_libm.so (the host's libm; shares nothing with the product):
    tests/test_bpots_oracle.py pins how often two libms that are each good to an ulp reach different -- equally valid --
    estimates.  The GPU must live in the same band against it, and satisfy the reference's own assertions
    (test/test_bpots.jl: the estimate reproduces the syndrome) as often as the libm build does."""
    def band(H, per, iters, T, C, S):
        M = sp.csc_matrix(H)
        M.sort_indices()
        dec = ldpc.BPOTSDecoder(M, per, iters, T=T, C=C)
        ga, _, _ = dec.decode_batch_host(S)
        dec.close()
        ob, _, _ = BPOTSOracle((M.indptr, M.indices), M.shape, per, iters, T, C, libm=True).batchdecode(S)
        Hd = np.asarray(M.todense()).astype(int)
        ok = lambda e: float(np.all((e.astype(int) @ Hd.T % 2) == (S & 1), axis=1).mean())
        return float((ga != ob).any(axis=1).mean()), ok(ga), ok(ob)

    rng = np.random.default_rng(3)
    for n, limit in ((4, 0.0), (8, 0.05), (16, 0.15)):
        H = cycle_matrix(n)
        for T, C in ((3, 1.0), (9, 3.0)):
            S = (rng.integers(0, 2, (300, n)).astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
            d, oa, ob = band(H, 0.01, 100, T, C, S)
            assert d <= limit and abs(oa - ob) <= 0.03, (n, T, C, d, oa, ob)
    H = toric_x(3)
    E = rng.integers(0, 2, (1000, 18)).astype(np.uint8)
    S = (E.astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
    d, oa, ob = band(H, 0.05, 50, 9, 3.0, S)
    assert d <= 0.30 and oa >= 0.85 and ob >= 0.85 and abs(oa - ob) <= 0.02, (d, oa, ob)
    H = ldpc.codes.parity_check_csc(504, 6, 3)
    S = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(504, 150, 0.09, seed=2))
    d, oa, ob = band(H, 0.09, 60, 9, 2.0, S)
    assert d <= 0.02 and abs(oa - ob) <= 0.02, (d, oa, ob)
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    S = ldpc.codes.syndromes_of(HX, ldpc.codes.random_errors(72, 1500, 0.04, seed=3))
    d, oa, ob = band(HX, 0.04, 50, 9, 2.0, S)
    assert d <= 0.01 and oa >= 0.99 and ob >= 0.99, (d, oa, ob)


def test_graphs_beyond_the_lds_take_the_node_kernel(ldpc, gpu):
    """Graphs whose messages do not fit a CU's LDS (round 1: LDPC_ERR_UNSUPPORTED) are decoded by bpots_node_kernel --
    one workgroup per syndrome, messages / LLRs / oscillation counters in a global slot.  The (4,8)-regular n = 16384
    code of BASELINE configs 3/4 and a (3,6) n = 8190 code, below and above threshold (the bias step runs), against the
    oracle: estimates, flags and iteration counts identical."""
    for n, wr, wc, per, B, iters in [(16384, 8, 4, 0.03, 40, 30), (16384, 8, 4, 0.08, 24, 20), (8190, 6, 3, 0.07, 60, 40)]:
        H = ldpc.codes.parity_check_csc(n, wr, wc)
        syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, per, seed=n + B))
        err, conv, its = assert_same_as_oracle(ldpc, H, per, iters, 9, 2.0, syn, kernel=3)
        if per <= 0.03:
            assert conv.all()
        else:
            assert (its >= 18).any()      # at least two bias steps somewhere


def test_node_kernel_equals_lds_kernel_on_small_graphs(ldpc, gpu, monkeypatch):
    """The same small graphs through both kernels (LDPC_BPOTS_FORCE_NODE sends everything to the node kernel): cycle
    matrices, toric code, irregular graphs with empty rows / columns and non-binary syndrome entries, BB-72."""
    monkeypatch.setenv("LDPC_BPOTS_FORCE_NODE", "1")
    rng = np.random.default_rng(12)
    for n in (4, 8, 16):
        H = cycle_matrix(n)
        S = (rng.integers(0, 2, (90, n)).astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
        assert_same_as_oracle(ldpc, H, 0.01, 100, 3, 1.0, S, kernel=3)
    H = toric_x(3)
    E = rng.integers(0, 2, (300, 18)).astype(np.uint8)
    assert_same_as_oracle(ldpc, H, 0.05, 50, 9, 3.0, (E.astype(int) @ H.T.astype(int) % 2).astype(np.uint8), kernel=3)
    for trial in range(3):
        H = (rng.random((20, 40)) < 0.12).astype(np.uint8)
        H[0, :] = 0
        H[:, 1] = 0
        syn = rng.integers(0, 2, (130, 20)).astype(np.uint8)
        syn[3, 2] = 2
        assert_same_as_oracle(ldpc, H, 0.03, 30, 4, 1.5, syn, kernel=3)
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    assert_same_as_oracle(ldpc, HX, 0.05, 40, 5, 2.0, ldpc.codes.syndromes_of(HX, ldpc.codes.random_errors(72, 700, 0.05, seed=2)), kernel=3)
