"""The host side of the team kernel's rows-in-LDS mode (bp_team_kernels.hpp TeamRows, ldpc_mi355x.hip
team_rows_tables()), checked without a GPU through ldpc_debug_team_rows: whatever the kernel reads from these tables
must describe the graph it decodes -- the dealt bit order is a permutation, every position carries its bit's four
message rows, a row lives in LDS only if its check AND its bit belong to the same member (nobody else may ever need it),
no two rows share an LDS slot, and the per-check and per-member views agree with the per-edge one."""
import ctypes

import numpy as np
import pytest

import ldpcdecoders_jl_amd as ldpc

RMAX = 312


def tables(n, members):
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    s = H.shape[0]
    colptr = np.ascontiguousarray(H.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(H.indices, dtype=np.int64)
    R = ctypes.c_int32()
    vtab = np.zeros((n, 16), dtype=np.int32)
    ctab = np.zeros((s, 2), dtype=np.int32)
    lds_edge = np.full(members * RMAX, -7, dtype=np.int32)
    st = ldpc._capi.lib().ldpc_debug_team_rows(s, n, colptr.ctypes.data, rowval.ctypes.data, members, ctypes.byref(R),
                                               vtab.ctypes.data, ctab.ctypes.data, lds_edge.ctypes.data)
    ldpc._capi.check(st)
    return H, R.value, vtab, ctab, lds_edge[: members * R.value].reshape(members, R.value)


@pytest.mark.parametrize("n,members", [(4096, 8), (16384, 32), (16384, 28), (1024, 3), (32768, 32)])
def test_row_tables_describe_the_graph(n, members):
    H, R, vtab, ctab, lds_edge = tables(n, members)
    s, nnz = H.shape[0], H.nnz
    assert 1 <= R <= RMAX
    # CSR row of every CSC edge, as ldpc_bp_create lays the rows out: 8 * check + place among the check's bits
    csr = H.tocsr()
    csr.sort_indices()
    place = {}
    for i in range(s):
        for k, j in enumerate(csr.indices[csr.indptr[i]:csr.indptr[i + 1]]):
            place[(i, int(j))] = 8 * i + k
    bits = vtab[:, 8] & 0x7FFFFFFF
    assert np.array_equal(np.sort(bits), np.arange(n)), "the dealt bit order must be a permutation of the bits"
    flagged = vtab[:, 8] < 0
    member_of_pos = (np.arange(n) // 4) % members                  # how the kernel deals positions: chunks of 4
    member_of_check = (np.arange(s) // 2) % members                # ... and checks: chunks of 2
    seen_slots = set()
    in_lds = 0
    for p in range(n):
        j = int(bits[p])
        rows = [place[(int(i), j)] for i in H.indices[H.indptr[j]:H.indptr[j + 1]]]
        assert list(vtab[p, 0:4]) == rows, "a position must carry the four message rows of its bit, checks ascending"
        lrows = vtab[p, 4:8]
        assert bool(flagged[p]) == bool((lrows >= 0).any())
        for q, lr in zip(rows, lrows):
            if lr < 0:
                continue
            in_lds += 1
            m = int(member_of_pos[p])
            assert member_of_check[q // 8] == m, "a row in LDS must belong to ONE member in both sweeps"
            assert 0 <= lr < R and (m, int(lr)) not in seen_slots
            seen_slots.add((m, int(lr)))
            assert lds_edge[m, lr] == q, "the member's write-back list must name the same row"
    assert in_lds == int((lds_edge >= 0).sum()) and np.all(lds_edge[lds_edge < 0] == -1)
    # per check: mask = its edges in LDS, base = LDS row of the first, the others follow
    for i in range(s):
        mask, base = int(ctab[i, 0]) & 0xFFFFFFFF, int(ctab[i, 1])
        m = int(member_of_check[i])
        want = [k for k in range(8) if (lds_edge[m] == 8 * i + k).any()]
        assert [k for k in range(8) if (mask >> k) & 1] == want
        for t, k in enumerate(want):
            assert lds_edge[m, base + t] == 8 * i + k
    # what it is for: about a quarter of the edges are candidates, the LDS holds up to 312 rows per member
    assert in_lds >= min(0.15 * nnz, 0.9 * RMAX * members) * 0.9
    per_member = (lds_edge >= 0).sum(axis=1)
    assert per_member.max() == R and per_member.min() >= 0.8 * min(R, nnz // 4 // members * 0.9)


def test_only_4_8_regular_graphs():
    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    colptr = np.ascontiguousarray(H.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(H.indices, dtype=np.int64)
    R = ctypes.c_int32()
    buf = np.zeros(1 << 20, dtype=np.int32)
    st = ldpc._capi.lib().ldpc_debug_team_rows(H.shape[0], 1008, colptr.ctypes.data, rowval.ctypes.data, 8, ctypes.byref(R),
                                               buf.ctypes.data, buf.ctypes.data, buf.ctypes.data)
    assert st == 5      # LDPC_ERR_UNSUPPORTED
