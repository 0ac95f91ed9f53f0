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


W = 8            # waves per member
RREGS = 32       # register rows a wave may hold (bp_team_kernels.hpp kTeamRegRows)


def tables(n, members, wr=8, wc=4, regs=0, quarters=3):
    H = ldpc.codes.parity_check_csc(n, wr, wc)
    s = H.shape[0]
    colptr = np.ascontiguousarray(H.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(H.indices, dtype=np.int64)
    deg = (ctypes.c_int32 * 2)()
    shape = (ctypes.c_int32 * 5)()
    vtab = np.zeros((n, 16), dtype=np.int32)
    ctab = np.zeros((s, 4), dtype=np.int32)
    lds_edge = np.full(members * RMAX, -7, dtype=np.int32)
    reg_edge = np.full(members * W * RREGS, -7, dtype=np.int32)
    st = ldpc._capi.lib().ldpc_debug_team_rows(s, n, colptr.ctypes.data, rowval.ctypes.data, members, regs, quarters,
                                               ctypes.byref(deg), ctypes.byref(shape), vtab.ctypes.data, ctab.ctypes.data,
                                               lds_edge.ctypes.data, reg_edge.ctypes.data)
    ldpc._capi.check(st)
    vt, R, static_c, static_v, regs_eff = list(shape)
    assert list(deg) == [wr, wc] and vt == (8 if 2 * wc + 1 <= 8 else 16)
    vtab = vtab.reshape(-1)[: n * vt].reshape(n, vt)
    reg_edge = reg_edge[: members * W * max(regs_eff, 1)].reshape(members, W, max(regs_eff, 1))
    return H, R, vtab, ctab, lds_edge[: members * R].reshape(members, R), reg_edge, (static_c, static_v, regs_eff)


@pytest.mark.parametrize("n,members,wr,wc,regs", [(4096, 8, 8, 4, 0), (16384, 32, 8, 4, 0), (16384, 32, 8, 4, 32), (16384, 28, 8, 4, 32), (1024, 3, 8, 4, 32),
                                                  (32768, 32, 8, 4, 32), (16380, 32, 6, 3, 32), (1008, 5, 6, 3, 0), (16000, 32, 10, 5, 32),
                                                  (4000, 7, 10, 5, 20),
                                                  # round 4: every regular pair of check degree 6 ... 10 x bit degree 3 ... 5 has an instantiation
                                                  (16380, 32, 9, 3, 32), (16380, 32, 10, 4, 32), (16380, 32, 7, 4, 32), (16002, 32, 9, 5, 32),
                                                  (16380, 32, 6, 5, 32), (3990, 6, 7, 3, 12), (16384, 32, 8, 3, 32)])
def test_row_tables_describe_the_graph(n, members, wr, wc, regs):
    H, R, vtab, ctab, lds_edge, reg_edge, (static_c, static_v, regs_eff) = tables(n, members, wr, wc, regs)
    s, nnz = H.shape[0], H.nnz
    assert 1 <= R <= RMAX and regs_eff in (0, regs)
    assert static_c % W == 0 and static_v % W == 0 and W <= static_c <= (s // 2) // members and W <= static_v <= (n // 4) // members
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

    def wave_of(chunk, static):                                    # the wave that owns a chunk by right, or -1 (dealt from the counter)
        l = chunk // members
        return l % W if l < static else -1

    seen_slots, seen_regs = set(), set()
    in_lds = in_regs = 0
    for p in range(n):
        j = int(bits[p])
        rows = [place[(int(i), j)] for i in H.indices[H.indptr[j]:H.indptr[j + 1]]]
        assert list(vtab[p, 0:wc]) == rows, "a position must carry the message rows of its bit, checks ascending"
        where = vtab[p, wc:2 * wc]
        assert bool(flagged[p]) == bool((where != -1).any())
        m = int(member_of_pos[p])
        for q, lr in zip(rows, where):
            if lr == -1:
                continue
            assert member_of_check[q // wr] == m, "a row on chip must belong to ONE member in both sweeps"
            if lr >= 0:
                in_lds += 1
                assert lr < R and (m, int(lr)) not in seen_slots
                seen_slots.add((m, int(lr)))
                assert lds_edge[m, lr] == q, "the member's write-back list must name the same row"
            else:
                in_regs += 1
                x = -2 - int(lr)
                w = wave_of(p // 4, static_v)
                assert regs_eff > 0 and 0 <= x < regs_eff and w >= 0, "a register row needs a position that a wave owns by right"
                assert wave_of((q // wr) // 2, static_c) == w, "... and a check that the SAME wave owns by right"
                assert (m, w, x) not in seen_regs
                seen_regs.add((m, w, x))
                assert reg_edge[m, w, x] == q, "the wave's write-back list must name the same row"
    assert in_lds == int((lds_edge >= 0).sum()) and np.all(lds_edge[lds_edge < 0] == -1)
    assert in_regs == int((reg_edge >= 0).sum()) and np.all(reg_edge[reg_edge < 0] == -1)
    # per check: masks = its edges in LDS / in registers, bases = the first row of either kind, the others follow
    for i in range(s):
        mask, base, rmask, rbase = int(ctab[i, 0]) & 0xFFFFFFFF, int(ctab[i, 1]), int(ctab[i, 2]) & 0xFFFFFFFF, int(ctab[i, 3])
        m = int(member_of_check[i])
        want = [k for k in range(wr) if (lds_edge[m] == wr * i + k).any()]
        assert [k for k in range(32) if (mask >> k) & 1] == want
        for t, k in enumerate(want):
            assert lds_edge[m, base + t] == wr * i + k
        w = wave_of(i // 2, static_c)
        want_r = [k for k in range(wr) if w >= 0 and (reg_edge[m, w] == wr * i + k).any()]
        assert [k for k in range(32) if (rmask >> k) & 1] == want_r and not (mask & rmask)
        for t, k in enumerate(want_r):
            assert reg_edge[m, w, rbase + t] == wr * i + k
    # what it is for: one edge per bit (1 / wc of the edges) is a candidate, the LDS holds up to 312 rows per member,
    # the registers of its eight waves up to 8 x 32 more
    assert in_lds + in_regs >= min(0.6 / wc * nnz, 0.9 * (RMAX + W * regs_eff) * members) * 0.9
    per_member = (lds_edge >= 0).sum(axis=1)
    assert per_member.max() == R
    if regs_eff and nnz // wc // members >= 400:
        assert in_regs >= 0.5 * regs_eff * W * members, "most register rows should be in use on a graph of this size"


@pytest.mark.parametrize("n,members,wr,wc", [(16384, 32, 8, 4), (16380, 23, 6, 3), (16000, 32, 10, 5), (16380, 32, 9, 3), (16380, 32, 10, 4), (16380, 32, 7, 5)])
def test_on_chip_rows_are_whole_checks_of_the_first_block(n, members, wr, wc):
    """What the fast paths of the kernel live on (bp_team_kernels.hpp check_update_regs, bit_update_pair_first /
    bit_update_multi_first; ldpc_mi355x.hip team_rows_tables(): a bit goes to the owner of its FIRST check): in a
    Gallager code the rows a member keeps on chip are its first-block checks, complete, as far as its capacity (312 rows
    in LDS, 8 x 32 in registers) goes -- a check with all its rows in one wave's registers or all in LDS needs no pointer
    per edge, and every bit has at most its first edge on chip plus a stray one now and then."""
    H, R, vtab, ctab, lds_edge, reg_edge, (static_c, static_v, regs_eff) = tables(n, members, wr, wc, regs=RREGS)
    s = H.shape[0]
    full = (1 << wr) - 1
    whole_regs = int((ctab[:, 2] == full).sum())
    whole_lds = int((ctab[:, 0] == full).sum())
    first_block = s // wc                                         # checks of the first block: wr consecutive bits each
    capacity = members * (RMAX + W * RREGS) // wr                  # whole checks the members can hold
    assert whole_regs + whole_lds >= 0.9 * min(first_block, capacity), (whole_regs, whole_lds, first_block, capacity)
    assert whole_regs >= 0.9 * min(first_block, members * W * (RREGS // wr)), whole_regs
    # whole checks on chip are first-block checks (check index < s / wc)
    assert int(np.flatnonzero((ctab[:, 0] | ctab[:, 2]) == full).max()) < first_block
    # every bit: edges 1 ... wc-1 in the slot except for strays (< 15 % of the bits); the first edge on chip for most
    loc = vtab[:, wc:2 * wc]
    rest_on_chip = (loc[:, 1:] != -1).any(axis=1)
    assert rest_on_chip.mean() < 0.15, rest_on_chip.mean()
    assert (loc[:, 0] != -1).mean() >= 0.9 * min(1.0, capacity / first_block), (loc[:, 0] != -1).mean()


def test_only_regular_graphs_with_an_instantiation():
    buf = np.zeros(1 << 20, dtype=np.int32)
    shape, deg = (ctypes.c_int32 * 5)(), (ctypes.c_int32 * 2)()

    def status(H):
        H = H.tocsc()
        H.sort_indices()
        colptr = np.ascontiguousarray(H.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(H.indices, dtype=np.int64)
        return ldpc._capi.lib().ldpc_debug_team_rows(H.shape[0], H.shape[1], colptr.ctypes.data, rowval.ctypes.data, 8, 32, 3,
                                                     ctypes.byref(deg), ctypes.byref(shape), buf.ctypes.data,
                                                     buf.ctypes.data, buf.ctypes.data, buf.ctypes.data)

    assert status(ldpc.codes.parity_check_csc(1008, 6, 3)) == 0
    assert status(ldpc.codes.parity_check_csc(1000, 10, 9)) == 5      # LDPC_ERR_UNSUPPORTED: no (10,9) instantiation
    H = ldpc.codes.parity_check_csc(1008, 6, 3).tolil()
    i, j = H.nonzero()
    H[i[0], j[0]] = 0                                                  # one edge less: not regular any more
    assert status(H.tocsc()) == 5


def test_stray_bits_share_position_chunks():
    """A bit with a LATER edge on chip (the room a member has beyond the whole checks of its block) takes its position
    chunk off the four-at-once update of the variable sweep (bp_team_kernels.hpp bit_update_multi_first wants every
    non-first edge in the slot).  team_rows_tables() gives such bits a member's -- and, among the bits with a row in a wave's
    registers, that wave's -- LAST positions, so they share chunks: at the C3 size 12 ... 18 chunks of a member's 128 (dealt
    by number they sat in 30 ... 52).  Check degree 10 keeps them dealt by number (measured: team_reg_plan())."""
    def stray_chunks(n, wr, wc):
        H, R, vtab, ctab, lds_edge, reg_edge, _ = tables(n, 32, wr, wc, RREGS)
        stray = (vtab[:, wc + 1:2 * wc] != -1).any(axis=1)
        per_chunk = stray[: n // 4 * 4].reshape(-1, 4)
        member = np.arange(n // 4) % 32
        return int(stray.sum()), [int(per_chunk[member == m].any(axis=1).sum()) for m in range(32)]

    total, chunks = stray_chunks(16384, 8, 4)
    assert total > 1000 and max(chunks) <= 20 and sum(chunks) <= total / 4 + 32 * (W + 1), (total, chunks)
    total10, chunks10 = stray_chunks(16380, 10, 4)
    assert sum(chunks10) > 0.6 * total10, (total10, chunks10)          # dealt by number: nearly one chunk per stray
    assert stray_chunks(16380, 6, 3)[0] == 0                           # bit degree 3: whole checks only, no strays at all


def plan(nnz, batch, cache_mib=240, max_iters=50, regular=1, dv=4):
    out = (ctypes.c_int32 * 6)()
    ldpc._capi.check(ldpc._capi.lib().ldpc_debug_team_plan(nnz, max_iters, batch, cache_mib, dv if regular else 0, ctypes.byref(out)))
    return dict(zip(("members", "teams", "grid", "xcds", "scatter", "rows"), list(out)))


def test_team_plan_by_graph_and_batch():
    """The host's choice of teams (ldpc_mi355x.hip team_plan_pure / team_fit), for an MI355X's geometry.  Message slot of
    a graph: nnz x 512 B; budget of slots in flight: 240 MiB of the 256 MiB Infinity Cache."""
    c3 = 65536                                                   # n = 16384, (4,8)-regular: 32 MiB a slot
    # the headline batch: EIGHT persistent teams of 32 -- with a quarter of a tile's rows on chip (312 in the LDS of every
    # member, 8 x 32 in its waves' registers) eight slots are 8 x 24 MiB of the 240 MiB budget.  A graph without a
    # rows-on-chip instantiation keeps whole slots: SEVEN teams (224 MiB; an eighth slot overfills the cache), 8 x 32
    # workgroups are launched and the blocks of the eighth XCD leave at once
    assert plan(c3, 65536) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)
    assert plan(c3, 65536, regular=0) == dict(members=32, teams=7, grid=256, xcds=7, scatter=0, rows=0)
    # exactly eight tiles: one round of eight teams rather than seven teams twice
    assert plan(c3, 512) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)
    assert plan(c3, 576)["teams"] == 8 and plan(c3, 576, regular=0)["teams"] == 7    # nine tiles: persistent again
    # up to three tiles: one team per tile, members dealt over all XCDs -- as many as leaves a member 512 message rows per
    # sweep, 192 at most (a single decode! of this code: 128 members 0.22 ms, 64 members 0.29 ms) --, no rows in LDS
    assert plan(c3, 1) == dict(members=128, teams=1, grid=128, xcds=8, scatter=1, rows=0)
    assert plan(c3, 192) == dict(members=85, teams=3, grid=255, xcds=8, scatter=1, rows=0)      # (one workgroup per CU in this plan)
    assert plan(c3, 256) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)      # four tiles: one round of teams, one per XCD
    assert plan(131072, 1)["members"] == 192 and plan(32768, 1)["members"] == 64
    # 28 MiB slots (n = 14336): eight fit the budget; the one team of an XCD takes ALL its 32 CUs as long as a member
    # keeps >= 1100 rows per sweep (1792 here), not only 28 of them (>= 2048 rows) -- all of them or none
    assert plan(57344, 65536)["teams"] == 8 and plan(57344, 65536)["members"] == 32
    assert plan(49152, 65536) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)      # n = 12288: 1536 rows
    assert plan(40960, 65536) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)      # n = 10240: 1280 rows
    assert plan(36864, 65536) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)      # n = 9216: 1152 rows
    # 16 MiB slots (n = 8192): members of >= 2048 rows are 16 at most, so two teams per XCD -- on seven XCDs with whole
    # slots (224 MiB), on all eight with the rows on chip taken off
    p = plan(32768, 65536, regular=0)
    assert p["members"] == 16 and p["teams"] == 14 and p["xcds"] == 7 and p["grid"] == 256
    p = plan(32768, 65536)
    assert p["members"] == 16 and p["teams"] == 16 and p["xcds"] == 8
    # the (3,6) n = 16380 code (24 MiB slots): eight teams of 32 (1535 rows a member; measured 528 ms for the full batch
    # against 633 ms with eight teams of 23 -- members of >= 2048 rows --, and twelve teams of 16 on six XCDs before that)
    assert plan(49140, 65536, dv=3) == dict(members=32, teams=8, grid=256, xcds=8, scatter=0, rows=1)
    # 64 MiB slots (n = 32768): eight of them are twice the cache.  With rows on chip (46 MiB a slot) FOUR fit: four WIDE
    # teams of 64 members each, dealt over all XCDs (round 4: 442 ms for 16,384 syndromes x 50 iterations against 496 ms
    # for one persistent team per XCD); a graph without rows on chip goes to the HBM-streaming tile kernel (an irregular
    # graph of that size: eight partly cached teams 570 ms, the tile kernel 518 ms), teams only while seven whole slots
    # are inside the budget (an irregular n = 20480 graph, 8 x 35 MiB: eight teams 381-392 ms, the tile kernel 319-328 ms)
    assert plan(131072, 65536) == dict(members=64, teams=4, grid=256, xcds=8, scatter=1, rows=1)
    assert plan(131072, 65536, regular=0)["members"] == 1
    assert plan(81920, 65536, regular=0)["members"] == 1 and plan(73728, 65536, regular=0)["members"] == 1
    assert plan(69632, 65536, regular=0) == dict(members=32, teams=7, grid=256, xcds=7, scatter=0, rows=0)      # 7 x 34 MiB
    # 128 MiB slots (n = 65536; 96 MiB with a quarter of the rows on chip): TWO wide teams of 128 (910 ms against the
    # tile kernel's 1238 ms); without rows on chip the tile kernel for more tiles than CUs, one team per tile below
    assert plan(262144, 65536) == dict(members=128, teams=2, grid=256, xcds=8, scatter=1, rows=1)
    assert plan(262144, 1024) == dict(members=128, teams=2, grid=256, xcds=8, scatter=1, rows=1)
    assert plan(262144, 65536, regular=0)["members"] == 1
    p = plan(262144, 1024, regular=0)                            # 16 tiles: two teams per XCD
    assert p["teams"] == 16 and p["members"] == 16 and p["scatter"] == 0
    # 256 MiB slots (n = 131072): ONE team of all 256 workgroups; beyond that (n = 262144) the tile kernel again
    assert plan(524288, 65536) == dict(members=256, teams=1, grid=256, xcds=8, scatter=1, rows=1)
    assert plan(1048576, 65536)["members"] == 1
    # slots that miss the budget by less than a quarter keep the one-XCD teams (n = 20480: 8 x 31 MiB)
    assert plan(81920, 65536)["scatter"] == 0
    # no budget at all (LDPC_TEAM_CACHE_MIB=0): round 1's rule
    assert plan(c3, 65536, cache_mib=0)["members"] == 1
    assert plan(c3, 2048, cache_mib=0) == dict(members=8, teams=32, grid=256, xcds=8, scatter=0, rows=0) or \
        plan(c3, 2048, cache_mib=0)["teams"] == 32
    # graphs too small for teams (fewer than 3 x 2048 rows) and batches whose mismatch words would not fit
    assert plan(4096, 65536)["members"] == 1
    assert plan(c3, 1 << 26, max_iters=4000)["members"] == 1


def _irr_tables(H, members, dcb=8, dvb=4):
    H = H.tocsc()
    H.sort_indices()
    s, n = H.shape
    colptr = np.ascontiguousarray(H.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(H.indices, dtype=np.int64)
    shape = (ctypes.c_int32 * 2)()
    ctab2 = np.zeros((s + 1, 2), dtype=np.int32); ptab = np.zeros((n + 1, 2), dtype=np.int32)
    ploc = np.zeros(H.nnz, dtype=np.int32); lds_edge = np.full(members * RMAX, -7, dtype=np.int32); posmap = np.zeros(n, dtype=np.int32)
    ldpc._capi.check(ldpc._capi.lib().ldpc_debug_team_irr(s, n, colptr.ctypes.data, rowval.ctypes.data, members, dcb, dvb, ctypes.byref(shape),
                                                          ctab2.ctypes.data, ptab.ctypes.data, ploc.ctypes.data, lds_edge.ctypes.data, posmap.ctypes.data))
    R, in_lds = list(shape)
    return H, R, in_lds, ctab2, ptab, ploc, lds_edge[: members * R].reshape(members, R), posmap


def _irregular_graph(n, s, seed, heavy=False):
    """Bits of degree 2 ... 5, checks of whatever degree that gives (about 6 ... 12), a few empty and heavy nodes."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for j in range(n):
        d = int(rng.integers(2, 6))
        for i in rng.choice(s, d, replace=False):
            rows.append(int(i)); cols.append(j)
    if heavy:
        for j in rng.choice(n, 20, replace=False):       # a check wider than any register bucket, a few wide bits
            rows.append(7); cols.append(int(j))
        for i in rng.choice(s, 19, replace=False):
            rows.append(int(i)); cols.append(11)
    H = sp.csc_matrix((np.ones(len(rows), dtype=np.uint8), (rows, cols)), shape=(s, n))
    H.data[:] = 1
    H.sum_duplicates()
    H.data[:] = 1
    return H


@pytest.mark.parametrize("n,s,members,heavy", [(4096, 2048, 8, False), (16384, 8192, 32, False), (16384, 8192, 32, True), (1000, 700, 3, True)])
def test_irregular_graph_tables_describe_the_graph(n, s, members, heavy):
    """Whole checks of an IRREGULAR graph in the LDS of their owners (ldpc_mi355x.hip team_irr_tables(); bp_team_kernels.hpp,
    IRR), checked without a GPU: the dealt bit order is a permutation; a position's edge list is its bit's (checks
    ascending), every entry either the CSR row itself or an LDS row of the member that owns the position; a check in LDS is
    there WHOLE, consecutively, with its owner, and every one of its bits sits in a position of that member (so nobody
    else ever touches those rows in either sweep); no two rows share an LDS slot; nodes wider than the register buckets
    stay in the slot; and the packing is worth having."""
    H0 = _irregular_graph(n, s, seed=n + members, heavy=heavy)
    # the kernel's register buckets for this graph's widest nodes (pickers.hpp): checks 8 / 16 / 32, bits 4 / 16
    maxc, maxv = int(np.diff(H0.tocsr().indptr).max()), int(np.diff(H0.tocsc().indptr).max())
    dcb, dvb = (8 if maxc <= 8 else 16 if maxc <= 16 else 32), (4 if maxv <= 4 else 16)
    H, R, in_lds, ctab2, ptab, ploc, lds_edge, posmap = _irr_tables(H0, members, dcb, dvb)
    nnz = H.nnz
    csr = H.tocsr(); csr.sort_indices()
    assert 1 <= R <= RMAX and np.array_equal(ctab2[:, 0], csr.indptr) and ptab[n, 0] == nnz
    bits = ptab[:n, 1] & 0x7FFFFFFF
    assert np.array_equal(np.sort(bits), np.arange(n)) and np.array_equal(posmap[bits], np.arange(n))
    place = {}
    for i in range(s):
        for k, j in enumerate(csr.indices[csr.indptr[i]:csr.indptr[i + 1]]):
            place[(i, int(j))] = int(csr.indptr[i]) + k
    member_of_pos = (np.arange(n) // 4) % members
    member_of_check = (np.arange(s) // 2) % members
    check_of_row = np.repeat(np.arange(s), np.diff(csr.indptr))
    seen = set()
    counted = 0
    for p in range(n):
        j = int(bits[p])
        rows = [place[(int(i), j)] for i in H.indices[H.indptr[j]:H.indptr[j + 1]]]
        loc = ploc[ptab[p, 0]:ptab[p + 1, 0]]
        assert len(loc) == len(rows)
        any_lds = False
        for q, l in zip(rows, loc):
            if l >= 0:
                assert l == q
                continue
            any_lds = True
            counted += 1
            i, m, lr = int(check_of_row[q]), int(member_of_pos[p]), -1 - int(l)
            assert member_of_check[i] == m, "a row in LDS must belong to ONE member in both sweeps"
            assert ctab2[i, 1] >= 0 and lr == ctab2[i, 1] + (q - csr.indptr[i]) and lr < R
            assert (m, lr) not in seen and lds_edge[m, lr] == q
            seen.add((m, lr))
            assert len(rows) <= dvb, "a bit wider than the register bucket never has a row in LDS"
        assert bool(ptab[p, 1] < 0) == any_lds
    assert counted == in_lds == int((lds_edge >= 0).sum()) and np.all(lds_edge[lds_edge < 0] == -1)
    for i in range(s):                         # a check in LDS is there whole; the others not at all
        deg = int(csr.indptr[i + 1] - csr.indptr[i])
        if ctab2[i, 1] >= 0:
            assert 0 < deg <= dcb and all(lds_edge[member_of_check[i], ctab2[i, 1] + k] == csr.indptr[i] + k for k in range(deg))
        else:
            assert not np.isin(np.arange(csr.indptr[i], csr.indptr[i + 1]), lds_edge).any() if deg and i % 97 == 0 else True
    # worth having: a random graph of mean bit degree 3.5 packs ~1 check in 4 (the bound is n / mean check degree), unless the
    # members' LDS is what limits it (312 rows each)
    assert in_lds >= min(0.08 * nnz, 0.7 * RMAX * members), (in_lds, nnz)
