"""Host logic that needs no GPU: code generators, PCM text I/O, syndrome normalisation, shards."""
import numpy as np
import pytest
import scipy.sparse as sp

import ldpcdecoders_jl_amd as ldpc
from ldpcdecoders_jl_amd import sharding  # noqa: F401  (import must work without a GPU)


def test_gallager_csc_matches_dense_and_is_sorted():
    for n, wr, wc in [(1008, 6, 3), (16384, 8, 4), (96, 6, 3)]:
        M = ldpc.codes.parity_check_csc(n, wr, wc)
        assert M.shape == (n * wc // wr, n) and M.nnz == n * wc
        assert np.all(np.diff(M.indptr) == wc)
        ind = M.indices.reshape(n, wc)
        assert np.all(np.diff(ind, axis=1) > 0)          # rows ascending inside every column
        assert np.all(np.asarray(M.sum(axis=1)).ravel() == wr)
    with pytest.raises(AssertionError):
        ldpc.codes.parity_check_csc(1000, 7, 3)          # @assert n % wr == 0 (parity_generator.jl:25)


def test_save_load_pcm_roundtrip(tmp_path):
    H = ldpc.parity_check_matrix(96, 6, 3)
    f = tmp_path / "H.pcm"
    ldpc.save_pcm(H, f)
    txt = f.read_text().splitlines()
    assert len(txt) == 48 and set(txt[0].split("\t")) <= {"0", "1"}   # writedlm(Int.(H)): tab-delimited
    H2 = ldpc.load_pcm(f)
    assert H2.dtype == np.int64 and np.array_equal(H2, H.astype(np.int64))


def test_bb72_code_is_a_css_pair():
    HX, HZ = ldpc.codes.bivariate_bicycle_72_12_6()
    assert HX.shape == HZ.shape == (36, 72)
    assert np.all(HX.sum(1) == 6) and np.all(HX.sum(0) == 3)
    assert not ((HX.astype(int) @ HZ.T.astype(int)) % 2).any()


def test_syndrome_bytes_alphabet():
    f = ldpc.syndrome_bytes
    assert f(np.array([True, False])).tolist() == [1, 0]
    assert f(np.array([0, 1, 2, 3, -1, -2])).tolist() == [0, 1, 2, 3, 3, 2]     # parity kept, never 0/1
    assert f(np.array([0.0, 1.0, 2.0])).tolist() == [0, 1, 2]
    with pytest.raises(ValueError):
        f(np.array([0.5]))
    with pytest.raises(TypeError):
        f(np.array(["a"]))


def test_syndromes_of_matches_dense():
    rng = np.random.default_rng(0)
    H = ldpc.parity_check_matrix(96, 6, 3)
    E = (rng.random((7, 96)) < 0.2).astype(np.uint8)
    S = ldpc.codes.syndromes_of(sp.csc_matrix(H), E)
    assert np.array_equal(S, (E.astype(int) @ H.T.astype(int)) % 2)


def test_shard_bounds_cover_the_batch():
    for B in [0, 1, 7, 64, 65536, 524288 + 3]:
        for G in [1, 2, 3, 8]:
            b = sharding.shard_bounds(B, G)
            assert b[0][0] == 0 and b[-1][1] == B
            assert all(b[i][1] == b[i + 1][0] for i in range(G - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
