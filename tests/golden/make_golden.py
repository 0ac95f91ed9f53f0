#!/usr/bin/env python3
"""Generates the golden fixtures in this directory (run from the repo root:
`python tests/golden/make_golden.py`).

PARITY UNPINNED: the reference is Julia, cannot run here and holds no golden vectors
(SURVEY.md 8c), so these vectors are produced by the C oracle (oracle/bp_oracle.c) and,
for the small cases, cross-checked bit for bit against the independent Python
restatement (oracle/bp_reference_py.py) before being written.  A Julia run of the
reference, if ever available, should be diffed against these files first: every array
needed to replay a case is stored (zero-based CSC of H, per, max_iters, syndromes).

Each .npz holds: shape [s, n], colptr, rowval, per, max_iters, syndromes [B][s] u8,
errors [B][n] u8, converged [B] u8, iters [B] i32, llr [B][n] f64 (= scratch.log_probabs).
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import ldpcdecoders_jl_amd as ldpc  # noqa: E402
from oracle import BPOracle  # noqa: E402
from oracle.bp_reference_py import DensePyBP  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def emit(name, H, per, max_iters, syn, cross_check):
    M = sp.csc_matrix(H)
    M.sort_indices()
    oc = BPOracle(csc=(M.indptr, M.indices), shape=M.shape, per=per, max_iters=max_iters)
    err, conv, llr, its = oc.batchdecode(syn)
    if cross_check:
        Hd = np.asarray(M.todense()).astype(int).tolist()
        for b in range(syn.shape[0]):
            py = DensePyBP(Hd, per, max_iters)
            perr, pconv = py.decode(syn[b].tolist())
            assert bool(conv[b]) == pconv and py.iters == its[b], (name, b)
            assert np.array_equal(np.asarray(perr, dtype=np.uint8), err[b]), (name, b)
            pl = np.asarray(py.log_probabs)
            fin = np.isfinite(pl)
            assert np.array_equal(llr[b][~fin], pl[~fin]) and np.allclose(llr[b][fin], pl[fin], rtol=1e-14, atol=0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), shape=np.array(M.shape, dtype=np.int64),
                        colptr=M.indptr.astype(np.int64), rowval=M.indices.astype(np.int64),
                        per=np.float64(per), max_iters=np.int64(max_iters), syndromes=syn.astype(np.uint8),
                        errors=err, converged=conv, iters=its, llr=llr)
    print(f"{name}: {M.shape} B={syn.shape[0]} converged={int(conv.sum())} iters={its.min()}..{its.max()}")


def main():
    rng = np.random.default_rng(20261004)
    # 1. BASELINE configs[0]/[1] code: (3,6)-regular n=1008, mostly convergent + a few hard ones
    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    E = np.concatenate([ldpc.codes.random_errors(1008, 24, 0.01, seed=1),
                        ldpc.codes.random_errors(1008, 8, 0.08, seed=2)])
    emit("c1_regular_3_6_n1008", H, 0.01, 50, ldpc.codes.syndromes_of(H, E), cross_check=False)
    # 2. the reference's test code (test/test_bp_decoder.jl:7): (9,10)-regular n=1000
    H = ldpc.codes.parity_check_csc(1000, 10, 9)
    E = ldpc.codes.random_errors(1000, 16, 0.01, seed=3)
    emit("ref_test_code_1000_10_9", H, 0.01, 100, ldpc.codes.syndromes_of(H, E), cross_check=False)
    # 3. BASELINE configs[4] code: BB [[72,12,6]] H_X
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    E = ldpc.codes.random_errors(72, 96, 0.03, seed=4)
    emit("bb72_hx", HX, 0.005, 50, ldpc.codes.syndromes_of(HX, E), cross_check=True)
    # 4. edge cases on a small irregular graph: degree-0/1 nodes, a heavy check and a heavy bit
    s, n = 40, 80
    Hs = (rng.random((s, n)) < 0.06).astype(np.uint8)
    Hs[2, :] = 0
    Hs[:, 3] = 0
    Hs[4, :] = 0
    Hs[4, 9] = 1                     # degree-1 check
    Hs[6, :36] = 1                   # degree >= 36 check
    Hs[8:26, 40] = 1                 # degree >= 18 bit
    for tag, per in [("p1e-12", 1e-12), ("p0.5", 0.5), ("p0.999", 0.999), ("p0.05", 0.05)]:
        E = (rng.random((20, n)) < 0.05).astype(np.uint8)
        syn = (E.astype(int) @ Hs.T.astype(int) % 2).astype(np.uint8)
        syn[12:] = rng.integers(0, 2, (8, s))     # arbitrary, mostly inconsistent syndromes
        syn[19, 0] = 2                             # an entry that can never match (:181)
        syn[18, 5] = 3
        emit("edge_irregular_" + tag, Hs, per, 12, syn, cross_check=True)
    # 5. max_iters = 0 and a zero syndrome
    emit("edge_max_iters_0", Hs, 0.05, 0, np.zeros((3, s), dtype=np.uint8), cross_check=True)


if __name__ == "__main__":
    main()
