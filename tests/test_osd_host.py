"""BP+OSD host step: the bit-packed product implementation
(`ldpc_osd_postprocess_batch`, host code, no GPU needed) against the literal dense oracle
(oracle/osd_oracle.c) on identical inputs, plus the reference's properties
(test/test_bposd_decoder.jl): OSD output ALWAYS satisfies the syndrome; exact recovery at
per = 0.01 for orders 0, 2..5."""
import numpy as np
import pytest

import ldpcdecoders_jl_amd as ldpc
from oracle import BPOracle, osd_oracle_postprocess


def _bp(H, per, iters, syn):
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=iters)
    return oc.batchdecode(syn)


def _check(H, per, iters, E, order, threads=0):
    syn = ldpc.codes.syndromes_of(H, E)
    err, conv, llr, _ = _bp(H, per, iters, syn)
    post = ldpc.OSDPostProcessor(H, order)
    out = post.postprocess(syn, err, llr, nthreads=threads)
    Hd = np.asarray(H.todense()).astype(np.uint8)
    for b in range(syn.shape[0]):
        ref = osd_oracle_postprocess(Hd, syn[b], err[b], llr[b], order)
        assert np.array_equal(out[b], ref), f"syndrome {b}: product OSD differs from the oracle"
    # syndrome consistency (test_bposd_decoder.jl:37-47, 59-61), rank-deficient H included
    assert np.array_equal(ldpc.codes.syndromes_of(H, out), syn)
    return out, conv


@pytest.mark.parametrize("order", [0, 2, 3, 5])
def test_exact_recovery_low_error_rate(order):
    """test_bposd_decoder.jl:6-34: per = 0.01 -> guess == err for orders 0 and 2..5."""
    H = ldpc.codes.parity_check_csc(1000, 10, 9)
    E = ldpc.codes.random_errors(1000, 6, 0.01, seed=order)
    out, conv = _check(H, 0.01, 100, E, order)
    assert np.array_equal(out, E)


@pytest.mark.parametrize("order", [0, 2])
def test_syndrome_consistency_when_bp_fails(order):
    """test_bposd_decoder.jl:37-47: per = 0.2, BP does not converge, OSD still matches the syndrome."""
    H = ldpc.codes.parity_check_csc(200, 10, 9) if order else ldpc.codes.parity_check_csc(1000, 10, 9)
    n = H.shape[1]
    E = ldpc.codes.random_errors(n, 5, 0.2, seed=20 + order)
    out, conv = _check(H, 0.2, 100 if order == 0 else 30, E, order)
    assert not conv.all()


@pytest.mark.parametrize("order", [0, 1, 4])
def test_bb72_rank_deficient(order):
    """BB [[72,12,6]] H_X has rank 30 < 36: elimination must cope with dependent rows."""
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    import scipy.sparse as sp

    H = sp.csc_matrix(HX)
    E = ldpc.codes.random_errors(72, 300, 0.06, seed=order)
    _check(H, 0.005, 20, E, order, threads=3)


def test_irregular_small_graphs_all_orders():
    rng = np.random.default_rng(3)
    import scipy.sparse as sp

    for trial in range(6):
        Hd = (rng.random((14, 30)) < 0.2).astype(np.uint8)
        Hd[0, :] = 0
        H = sp.csc_matrix(Hd)
        E = (rng.random((12, 30)) < 0.15).astype(np.uint8)
        for order in (0, 1, 3, 7):
            _check(H, 0.1, 10, E, order)


def test_order_clamped_to_information_set():
    """osd_order > n - rank is clamped (the reference @warns, :174-177)."""
    import scipy.sparse as sp

    Hd = np.eye(5, 7, dtype=np.uint8)
    Hd[:, 5] = 1
    H = sp.csc_matrix(Hd)
    E = np.zeros((3, 7), dtype=np.uint8)
    E[0, 5] = 1
    E[1, 0] = E[1, 6] = 1
    _check(H, 0.1, 5, E, 6)


def test_non_binary_syndrome_rejected():
    H = ldpc.codes.parity_check_csc(96, 6, 3)
    post = ldpc.OSDPostProcessor(H, 0)
    syn = np.zeros((1, 48), dtype=np.uint8)
    syn[0, 0] = 2
    with pytest.raises(ldpc.LdpcError):
        post.postprocess(syn, np.zeros((1, 96), np.uint8), np.zeros((1, 96)))


def test_osd_host_under_sanitizers(tmp_path):
    """osd_host.cpp built with AddressSanitizer + UBSan (CPU only; the GPU pool offers no sanitizers) and
    driven by tests/native/osd_sanitize.cpp: ~8.6k syndrome x order x thread cases against the dense
    oracle, rank-deficient matrices and reliability ties included."""
    import os
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    obj = str(tmp_path / "osd_oracle.o")
    subprocess.check_call(["gcc", "-std=c11", "-ffp-contract=off", *san, "-c", os.path.join(root, "oracle", "osd_oracle.c"), "-o", obj])
    exe = str(tmp_path / "osd_sanitize")
    subprocess.check_call(["g++", "-std=c++17", *san, "-o", exe, os.path.join(root, "tests", "native", "osd_sanitize.cpp"),
                           os.path.join(root, "ldpcdecoders.jl_amd", "csrc", "osd_host.cpp"), obj, "-lpthread", "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
