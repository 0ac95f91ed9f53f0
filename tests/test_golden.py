"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle must reproduce them (and the independent Python restatement must on the
small cases).  GPU: the HIP path must reproduce them through the C ABI -- bit-exact hard
decisions / flags / iteration counts, LLRs within 1e-5, +-Inf exactly."""
import glob
import os

import numpy as np
import pytest

import ldpcdecoders_jl_amd as ldpc
from oracle import BPOracle
from oracle.bp_reference_py import DensePyBP

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _load(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def _check_llr(llr, ref, tol):
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(llr), fin)
    assert np.array_equal(llr[~fin], ref[~fin])
    if fin.any():
        assert np.max(np.abs(llr[fin] - ref[fin])) <= tol


def test_fixtures_exist():
    assert len(FIXTURES) >= 8


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(path):
    g = _load(path)
    s, n = (int(v) for v in g["shape"])
    oc = BPOracle(csc=(g["colptr"], g["rowval"]), shape=(s, n), per=float(g["per"]), max_iters=int(g["max_iters"]))
    err, conv, llr, its = oc.batchdecode(g["syndromes"])
    assert np.array_equal(err, g["errors"]) and np.array_equal(conv, g["converged"])
    assert np.array_equal(its, g["iters"])
    _check_llr(llr, g["llr"], 0.0)
    if n <= 100:  # pure-Python loops: small cases only
        import scipy.sparse as sp

        H = np.asarray(sp.csc_matrix((np.ones(len(g["rowval"])), g["rowval"], g["colptr"]), shape=(s, n)).todense())
        for b in range(0, g["syndromes"].shape[0], 3):
            py = DensePyBP(H.astype(int).tolist(), float(g["per"]), int(g["max_iters"]))
            perr, pconv = py.decode(g["syndromes"][b].tolist())
            assert pconv == bool(g["converged"][b]) and py.iters == g["iters"][b]
            assert np.array_equal(np.asarray(perr, dtype=np.uint8), g["errors"][b])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 0], ids=["hbm_streaming", "auto_lds"])
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_path_reproduces_golden(path, variant, gpu):
    import scipy.sparse as sp

    g = _load(path)
    s, n = (int(v) for v in g["shape"])
    H = sp.csc_matrix((np.ones(len(g["rowval"]), dtype=bool), g["rowval"], g["colptr"]), shape=(s, n))
    dec = ldpc.BeliefPropagationDecoder(H, float(g["per"]), int(g["max_iters"]), kernel_variant=variant)
    err, conv, llr, its = dec.decode_batch_host(g["syndromes"], want_llr=True, want_iters=True)
    assert np.array_equal(err, g["errors"])
    assert np.array_equal(conv, g["converged"])
    assert np.array_equal(its, g["iters"])
    _check_llr(llr, g["llr"], 1e-5)
