"""The C oracle against the independently written pure-Python restatement.

Both follow src/decoders/belief_propagation.jl:121-188; they share no code.
Everything but `log` must agree bit for bit (messages included)."""
import math

import numpy as np
import pytest

from oracle import BPOracle
from oracle.bp_reference_py import DensePyBP


def _rand_H(rng, s, n, density):
    return (rng.random((s, n)) < density).astype(np.uint8)


def _bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


CASES = [
    # (s, n, density, per, max_iters, error_rate)
    (6, 12, 0.4, 0.05, 10, 0.05),
    (12, 24, 0.25, 0.01, 20, 0.02),
    (20, 30, 0.2, 0.1, 15, 0.1),
    (15, 40, 0.15, 0.3, 8, 0.3),
    (10, 10, 0.5, 0.5, 6, 0.5),
    (8, 16, 0.3, 0.999, 5, 0.2),
    (8, 16, 0.3, 1e-12, 5, 0.1),
    (8, 16, 0.3, 0.0, 4, 0.1),
    (8, 16, 0.3, 1.0, 4, 0.1),
]


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_c_oracle_equals_python_restatement(case, dense):
    s, n, dens, per, iters, er = case
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    for trial in range(4):
        H = _rand_H(rng, s, n, dens)
        if trial == 1:
            H[0, :] = 0          # degree-0 check
            H[:, 0] = 0          # degree-0 bit
        e = (rng.random(n) < er).astype(np.uint8)
        syn = (H.astype(int) @ e % 2).astype(np.uint8)
        if trial == 2:
            syn = rng.integers(0, 2, s).astype(np.uint8)  # arbitrary (maybe inconsistent) syndrome
        py = DensePyBP(H.tolist(), per, iters)
        perr, pconv = py.decode(syn.tolist())
        oc = BPOracle(H, per, iters, dense=dense)
        cerr, cconv = oc.decode(syn)
        assert cconv == pconv
        assert oc.last_iters == py.iters
        assert np.array_equal(cerr, np.asarray(perr))
        cb, cc = oc.messages()
        pb, pc = py.messages_csc()
        assert np.array_equal(_bits(cb), _bits(pb)) or _nan_equal(cb, pb)
        assert np.array_equal(_bits(cc), _bits(pc)) or _nan_equal(cc, pc)
        cl, pl = oc.log_probabs, np.asarray(py.log_probabs)
        fin = np.isfinite(pl)
        assert np.array_equal(np.isfinite(cl), fin)
        assert np.array_equal(cl[~fin], pl[~fin])
        assert np.all(np.abs(cl[fin] - pl[fin]) <= 4 * np.spacing(np.abs(pl[fin])) + 0.0)


def _nan_equal(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint64), b[~nb].view(np.uint64))


def test_max_iters_zero():
    H = np.eye(4, 8, dtype=np.uint8) | np.eye(4, 8, k=2, dtype=np.uint8)
    oc = BPOracle(H, 0.1, 0)
    err, conv = oc.decode(np.zeros(4, dtype=np.uint8))
    assert not conv and not err.any() and not oc.log_probabs.any() and oc.last_iters == 0


def test_non_binary_syndrome_never_converges():
    # an Int 2 in the reference has sign (+1) but can never equal a parity (belief_propagation.jl:136,181)
    H = np.eye(4, 8, dtype=np.uint8) | np.eye(4, 8, k=2, dtype=np.uint8)
    oc = BPOracle(H, 0.1, 7)
    err0, conv0 = oc.decode(np.zeros(4, dtype=np.uint8))
    assert conv0 and oc.last_iters == 1
    err2, conv2 = oc.decode(np.array([2, 0, 0, 0], dtype=np.uint8))
    assert not conv2 and oc.last_iters == 7
    py = DensePyBP(H.tolist(), 0.1, 7)
    perr, pconv = py.decode([2, 0, 0, 0])
    assert not pconv and np.array_equal(err2, np.asarray(perr))


def test_oracles_under_sanitizers(tmp_path):
    """The C oracles themselves under AddressSanitizer + UBSan (tests/native/oracle_sanitize.c): random
    irregular graphs, per from 0 to 1, non-binary syndromes; edge-list and dense storage agree bit for bit."""
    import os
    import shutil
    import subprocess

    import pytest

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", os.path.join(root, "ldpcdecoders.jl_amd", "csrc"), "-o", exe,
                           os.path.join(root, "tests", "native", "oracle_sanitize.c"), os.path.join(root, "oracle", "bp_oracle.c"),
                           os.path.join(root, "oracle", "bpots_oracle.c"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
