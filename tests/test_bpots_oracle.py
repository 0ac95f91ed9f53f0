"""BP-OTS on the CPU: the C oracle against the independently written Python
restatement (bit-exact: both use the portable tanh/atanh), the portable math against libm, and the
reference's own assertions (test/test_bpots.jl: the estimate always reproduces the syndrome on the
cycle matrices, >= 85 % on a d=3 toric code)."""
import ctypes
import math

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import BPOTSOracle
from oracle.bpots import _lib
from oracle.bpots_reference_py import DensePyBPOTS


def cycle_matrix(n):                       # test_bpots.jl:13-24
    H = np.zeros((n, n), dtype=np.uint8)
    for j in range(n):
        H[j, (j + 1) % n] = 1
        H[j, j] = 1
    return H


def toric_x(d):
    """X-check matrix of the d x d toric code: 2 d^2 qubits (horizontal, vertical edges), d^2 vertex checks."""
    H = np.zeros((d * d, 2 * d * d), dtype=np.uint8)
    for r in range(d):
        for c in range(d):
            v = r * d + c
            H[v, r * d + c] = 1                          # horizontal edge to the right
            H[v, r * d + (c - 1) % d] = 1                # horizontal edge from the left
            H[v, d * d + r * d + c] = 1                  # vertical edge down
            H[v, d * d + ((r - 1) % d) * d + c] = 1      # vertical edge from above
    return H


def _oracle(H, per, iters, T, C):
    M = sp.csc_matrix(H)
    M.sort_indices()
    return BPOTSOracle((M.indptr, M.indices), M.shape, per, iters, T, C)


def test_portable_math_is_close_to_libm():
    L = _lib()
    for name, ref, xs in [("pm_tanh_export", math.tanh, np.concatenate([np.linspace(-25, 25, 20001), np.geomspace(1e-12, 0.3, 3000)])),
                          ("pm_atanh_export", math.atanh, np.concatenate([np.linspace(-0.99999, 0.99999, 20001), np.geomspace(1e-12, 0.3, 3000)])),
                          ("pm_exp_export", math.exp, np.linspace(-40, 40, 20001)),
                          ("pm_log_export", math.log, np.geomspace(1e-300, 1e300, 20001))]:
        f = getattr(L, name)
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_double]
        worst = 0.0
        for x in xs:
            a, b = f(float(x)), ref(float(x))
            worst = max(worst, abs(a - b) / max(np.spacing(abs(b)), 5e-324))
        assert worst <= 4.0, (name, worst)
    f = L.pm_atanh_export
    assert f(1.0) == math.inf and f(-1.0) == -math.inf and math.isnan(f(1.5)) and f(0.0) == 0.0


@pytest.mark.parametrize("n", [4, 8, 16])
@pytest.mark.parametrize("T,C", [(3, 1.0), (5, 2.0), (9, 3.0)])
def test_trapping_set_resistance(n, T, C):
    """test_bpots.jl:55-84: weight-2 error on the cycle matrix must be decoded to the syndrome."""
    H = cycle_matrix(n)
    e = np.zeros(n, dtype=np.uint8)
    e[:2] = 1
    syn = (H.astype(int) @ e % 2).astype(np.uint8)
    err, conv, its = _oracle(H, 0.01, 100, T, C).batchdecode(syn[None, :])
    assert np.array_equal(H.astype(int) @ err[0] % 2, syn)
    py = DensePyBPOTS(H.tolist(), 0.01, 100, T, C)
    perr, pconv = py.decode(syn.tolist())
    assert perr == err[0].tolist() and pconv == bool(conv[0]) and py.iters == its[0]


def test_c_oracle_equals_python_restatement_on_random_cases():
    rng = np.random.default_rng(5)
    for trial in range(12):
        s, n = int(rng.integers(4, 12)), int(rng.integers(8, 20))
        H = (rng.random((s, n)) < 0.3).astype(np.uint8)
        if trial % 3 == 0:
            H[0, :] = 0
        syn = rng.integers(0, 2, (6, s)).astype(np.uint8)
        if trial == 5:
            syn[0, 1] = 2            # non-zero flips the sign (:195), can never be matched (:273)
        per, iters, T, C = float(rng.choice([0.01, 0.05, 0.2])), 40, int(rng.choice([3, 5, 9])), float(rng.choice([1.0, 2.0, 3.0]))
        err, conv, its = _oracle(H, per, iters, T, C).batchdecode(syn)
        for b in range(6):
            py = DensePyBPOTS(H.tolist(), per, iters, T, C)
            perr, pconv = py.decode(syn[b].tolist())
            assert perr == err[b].tolist() and pconv == bool(conv[b]) and py.iters == its[b], (trial, b)


def test_toric_code_performance():
    """test_bpots.jl:116-137: d=3 toric code, random syndromes, >= 85 % reproduced at each noise level."""
    H = toric_x(3)
    assert H.shape == (9, 18) and np.all(H.sum(1) == 4) and np.all(H.sum(0) == 2)
    rng = np.random.default_rng(7)
    for noise in (0.01, 0.05, 0.1):
        E = rng.integers(0, 2, (100, 18)).astype(np.uint8)          # rand(Bool, n) :28
        syn = (E.astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
        err, conv, _ = _oracle(H, noise, 50, 9, 3.0).batchdecode(syn)
        ok = np.all((err.astype(int) @ H.T.astype(int) % 2) == syn, axis=1)
        assert ok.mean() >= 0.85


def _band(H, per, iters, T, C, S):
    """(fraction of syndromes on which the portable-math and the libm build of the oracle differ in estimate, flag or
    iteration count; share of syndromes whose estimate reproduces the syndrome: portable build, libm build)"""
    M = sp.csc_matrix(H)
    M.sort_indices()
    a = BPOTSOracle((M.indptr, M.indices), M.shape, per, iters, T, C).batchdecode(S)
    b = BPOTSOracle((M.indptr, M.indices), M.shape, per, iters, T, C, libm=True).batchdecode(S)
    differ = (a[0] != b[0]).any(axis=1) | (a[1] != b[1]) | (a[2] != b[2])
    Hi = np.asarray(M.todense()).astype(int)
    ok_a = np.all((a[0].astype(int) @ Hi.T % 2) == S, axis=1)
    ok_b = np.all((b[0].astype(int) @ Hi.T % 2) == S, axis=1)
    return differ.mean(), ok_a.mean(), ok_b.mean()


def test_independent_libm_oracle_and_the_band_two_correct_libms_live_in():
    """oracle/libbpots_oracle_libm.so is the same restatement with the HOST's tanh / atanh: it shares no arithmetic with
    the product (the default build shares portable_math.h with the HIP kernel, so GPU == oracle is by construction on
    the one part of BP-OTS where implementations legitimately differ).  BP-OTS takes data-dependent decisions on values
    that went through tanh / atanh, so two libms that are each within an ulp reach different -- equally valid -- estimates
    on a share of the syndromes; Julia's own libm is a third such pair.  This test (1) holds the reference's assertions
    (test/test_bpots.jl: the estimate reproduces the syndrome on the small cycle matrices; >= 85 % on the d = 3 toric
    code) against BOTH builds, and (2) pins the size of the band: zero to a few per mille on generic graphs, up to a
    fifth on the highly symmetric toric code, where degenerate solutions abound -- and both builds reproduce the
    syndrome equally often everywhere."""
    rng = np.random.default_rng(3)
    for n, limit in ((4, 0.0), (8, 0.05), (16, 0.15)):
        H = cycle_matrix(n)
        for T, C in ((3, 1.0), (5, 2.0), (9, 3.0)):
            S = (rng.integers(0, 2, (300, n)).astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
            d, oa, ob = _band(H, 0.01, 100, T, C, S)
            assert d <= limit and abs(oa - ob) <= 0.03, (n, T, C, d, oa, ob)
            if n <= 8:
                assert oa == 1.0 and ob == 1.0
    H = toric_x(3)
    for noise in (0.01, 0.05, 0.1):
        E = rng.integers(0, 2, (1000, 18)).astype(np.uint8)
        S = (E.astype(int) @ H.T.astype(int) % 2).astype(np.uint8)
        d, oa, ob = _band(H, noise, 50, 9, 3.0, S)
        assert 0.02 <= d <= 0.30 and oa >= 0.85 and ob >= 0.85 and abs(oa - ob) <= 0.02, (noise, d, oa, ob)
    # generic graphs: a regular LDPC code above its threshold, and the BB [[72,12,6]] check matrix
    import ldpcdecoders_jl_amd as ldpc

    H = ldpc.codes.parity_check_csc(504, 6, 3)
    S = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(504, 150, 0.09, seed=2))
    d, oa, ob = _band(H, 0.09, 60, 9, 2.0, S)
    assert d <= 0.02 and abs(oa - ob) <= 0.02
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    S = ldpc.codes.syndromes_of(HX, ldpc.codes.random_errors(72, 1500, 0.04, seed=3))
    d, oa, ob = _band(HX, 0.04, 50, 9, 2.0, S)
    assert d <= 0.01 and oa >= 0.99 and ob >= 0.99
