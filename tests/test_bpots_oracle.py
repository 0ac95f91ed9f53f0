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
