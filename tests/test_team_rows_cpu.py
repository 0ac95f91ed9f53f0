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


def tables(n, members, wr=8, wc=4):
    H = ldpc.codes.parity_check_csc(n, wr, wc)
    s = H.shape[0]
    colptr = np.ascontiguousarray(H.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(H.indices, dtype=np.int64)
    R, vt = ctypes.c_int32(), ctypes.c_int32()
    deg = (ctypes.c_int32 * 2)()
    vtab = np.zeros((n, 16), dtype=np.int32)
    ctab = np.zeros((s, 2), dtype=np.int32)
    lds_edge = np.full(members * RMAX, -7, dtype=np.int32)
    st = ldpc._capi.lib().ldpc_debug_team_rows(s, n, colptr.ctypes.data, rowval.ctypes.data, members, ctypes.byref(deg),
                                               ctypes.byref(vt), ctypes.byref(R), vtab.ctypes.data, ctab.ctypes.data,
                                               lds_edge.ctypes.data)
    ldpc._capi.check(st)
    assert list(deg) == [wr, wc] and vt.value == (8 if 2 * wc + 1 <= 8 else 16)
    vtab = vtab.reshape(-1)[: n * vt.value].reshape(n, vt.value)
    return H, R.value, vtab, ctab, lds_edge[: members * R.value].reshape(members, R.value)


@pytest.mark.parametrize("n,members,wr,wc", [(4096, 8, 8, 4), (16384, 32, 8, 4), (16384, 28, 8, 4), (1024, 3, 8, 4), (32768, 32, 8, 4),
                                             (16380, 32, 6, 3), (1008, 5, 6, 3), (16000, 32, 10, 5), (4000, 7, 10, 5)])
def test_row_tables_describe_the_graph(n, members, wr, wc):
    H, R, vtab, ctab, lds_edge = tables(n, members, wr, wc)
    s, nnz = H.shape[0], H.nnz
    assert 1 <= R <= RMAX
    # CSR row of every CSC edge, as ldpc_bp_create lays the rows out: wr * check + place among the check's bits
    csr = H.tocsr()
    csr.sort_indices()
    place = {}
    for i in range(s):
        for k, j in enumerate(csr.indices[csr.indptr[i]:csr.indptr[i + 1]]):
            place[(i, int(j))] = wr * i + k
    bits = vtab[:, 2 * wc] & 0x7FFFFFFF
    assert np.array_equal(np.sort(bits), np.arange(n)), "the dealt bit order must be a permutation of the bits"
    flagged = vtab[:, 2 * wc] < 0
    member_of_pos = (np.arange(n) // 4) % members                  # how the kernel deals positions: chunks of 4
    member_of_check = (np.arange(s) // 2) % members                # ... and checks: chunks of 2
    seen_slots = set()
    in_lds = 0
    for p in range(n):
        j = int(bits[p])
        rows = [place[(int(i), j)] for i in H.indices[H.indptr[j]:H.indptr[j + 1]]]
        assert list(vtab[p, 0:wc]) == rows, "a position must carry the message rows of its bit, checks ascending"
        lrows = vtab[p, wc:2 * wc]
        assert bool(flagged[p]) == bool((lrows >= 0).any())
        for q, lr in zip(rows, lrows):
            if lr < 0:
                continue
            in_lds += 1
            m = int(member_of_pos[p])
            assert member_of_check[q // wr] == m, "a row in LDS must belong to ONE member in both sweeps"
            assert 0 <= lr < R and (m, int(lr)) not in seen_slots
            seen_slots.add((m, int(lr)))
            assert lds_edge[m, lr] == q, "the member's write-back list must name the same row"
    assert in_lds == int((lds_edge >= 0).sum()) and np.all(lds_edge[lds_edge < 0] == -1)
    # per check: mask = its edges in LDS, base = LDS row of the first, the others follow
    for i in range(s):
        mask, base = int(ctab[i, 0]) & 0xFFFFFFFF, int(ctab[i, 1])
        m = int(member_of_check[i])
        want = [k for k in range(wr) if (lds_edge[m] == wr * i + k).any()]
        assert [k for k in range(32) if (mask >> k) & 1] == want
        for t, k in enumerate(want):
            assert lds_edge[m, base + t] == wr * i + k
    # what it is for: one edge per bit (1 / wc of the edges) is a candidate, the LDS holds up to 312 rows per member
    assert in_lds >= min(0.6 / wc * nnz, 0.9 * RMAX * members) * 0.9
    per_member = (lds_edge >= 0).sum(axis=1)
    assert per_member.max() == R and per_member.min() >= 0.8 * min(R, nnz // wc // members * 0.9)


def test_only_regular_graphs_with_an_instantiation():
    buf = np.zeros(1 << 20, dtype=np.int32)
    R, vt, deg = ctypes.c_int32(), ctypes.c_int32(), (ctypes.c_int32 * 2)()

    def status(H):
        H = H.tocsc()
        H.sort_indices()
        colptr = np.ascontiguousarray(H.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(H.indices, dtype=np.int64)
        return ldpc._capi.lib().ldpc_debug_team_rows(H.shape[0], H.shape[1], colptr.ctypes.data, rowval.ctypes.data, 8,
                                                     ctypes.byref(deg), ctypes.byref(vt), ctypes.byref(R), buf.ctypes.data,
                                                     buf.ctypes.data, buf.ctypes.data)

    assert status(ldpc.codes.parity_check_csc(1008, 6, 3)) == 0
    assert status(ldpc.codes.parity_check_csc(1000, 10, 9)) == 5      # LDPC_ERR_UNSUPPORTED: no (10,9) instantiation
    H = ldpc.codes.parity_check_csc(1008, 6, 3).tolil()
    i, j = H.nonzero()
    H[i[0], j[0]] = 0                                                  # one edge less: not regular any more
    assert status(H.tocsc()) == 5


def plan(nnz, batch, cache_mib=240, max_iters=50, regular=1, dv=4):
    out = (ctypes.c_int32 * 6)()
    ldpc._capi.check(ldpc._capi.lib().ldpc_debug_team_plan(nnz, max_iters, batch, cache_mib, dv if regular else 0, ctypes.byref(out)))
    return dict(zip(("members", "teams", "grid", "xcds", "scatter", "rows"), list(out)))


def test_team_plan_by_graph_and_batch():
    """The host's choice of teams (ldpc_mi355x.hip team_plan_pure / team_fit), for an MI355X's geometry.  Message slot of
    a graph: nnz x 512 B; budget of slots in flight: 240 MiB of the 256 MiB Infinity Cache."""
    c3 = 65536                                                   # n = 16384, (4,8)-regular: 32 MiB a slot
    # the headline batch: SEVEN persistent teams of 32 (224 MiB; an eighth slot overfills the cache), rows in LDS;
    # 8 x 32 workgroups are launched, the blocks of the eighth XCD leave at once
    assert plan(c3, 65536) == dict(members=32, teams=7, grid=256, xcds=7, scatter=0, rows=1)
    assert plan(c3, 65536, regular=0)["rows"] == 0
    # exactly eight tiles: one round of eight teams rather than seven teams twice
    assert plan(c3, 512) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)
    assert plan(c3, 576)["teams"] == 7                           # nine tiles: persistent again
    # up to four tiles: one team per tile, members dealt over all XCDs, up to 64 of them, no rows in LDS
    assert plan(c3, 1) == dict(members=64, teams=1, grid=64, xcds=8, scatter=1, rows=0)
    assert plan(c3, 256) == dict(members=64, teams=4, grid=256, xcds=8, scatter=1, rows=0)
    # 28 MiB slots (n = 14336): eight fit the budget
    assert plan(57344, 65536)["teams"] == 8 and plan(57344, 65536)["members"] == 28
    # 16 MiB slots (n = 8192): members of >= 2048 rows are 16 at most, so two teams per XCD on seven XCDs (224 MiB)
    p = plan(32768, 65536)
    assert p["members"] == 16 and p["teams"] == 14 and p["xcds"] == 7 and p["grid"] == 256
    # 64 MiB slots (n = 32768): twice the cache -- the second tier, one persistent team per XCD
    assert plan(131072, 65536) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)
    # 128 MiB slots (n = 65536): the tile kernel for more tiles than CUs, one team per tile below
    assert plan(262144, 65536)["members"] == 1
    p = plan(262144, 1024)                                       # 16 tiles: two teams per XCD
    assert p["teams"] == 16 and p["members"] == 16 and p["scatter"] == 0
    # no budget at all (LDPC_TEAM_CACHE_MIB=0): round 1's rule
    assert plan(c3, 65536, cache_mib=0)["members"] == 1
    assert plan(c3, 2048, cache_mib=0) == dict(members=8, teams=32, grid=256, xcds=8, scatter=0, rows=0) or \
        plan(c3, 2048, cache_mib=0)["teams"] == 32
    # graphs too small for teams (fewer than 3 x 2048 rows) and batches whose mismatch words would not fit
    assert plan(4096, 65536)["members"] == 1
    assert plan(c3, 1 << 26, max_iters=4000)["members"] == 1
