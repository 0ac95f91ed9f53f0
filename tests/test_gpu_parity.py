"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): hard decisions and convergence flags bit-exact,
iteration counts equal, LLRs (`scratch.log_probabs`) within 1e-5 with +-Inf
matching exactly.  Every test here runs the kernels on a real MI355X."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import BPOracle

pytestmark = pytest.mark.gpu

LLR_TOL = 1e-5  # north_star: "LLRs within 1e-5"


def assert_parity(ldpc, H, per, max_iters, syn_bs, **kw):
    """Decode syn_bs [B][s] on the GPU and with the oracle; compare everything."""
    M = sp.csc_matrix(H)
    M.sort_indices()
    oc = BPOracle(csc=(M.indptr, M.indices), shape=M.shape, per=per, max_iters=max_iters)
    oerr, oconv, ollr, oits = oc.batchdecode(syn_bs, want_llr=True)
    # all four kernels: 1 = HBM-streaming tile kernel, 0 = auto (the LDS-resident kernel whenever the
    # edge messages fit the LDS, which is the case for every small code used here; the node-parallel
    # kernel for small batches on larger codes, the team kernel for medium ones), 3 = node-parallel
    # kernel (one workgroup per syndrome), 4 = team kernel (several workgroups per 64-syndrome tile)
    default = [1, 0, 3] if ("waves_per_tile" in kw or "resident_tiles" in kw) else [1, 0, 3, 4]
    for variant in ([kw.pop("kernel_variant")] if "kernel_variant" in kw else default):
        dec = ldpc.BeliefPropagationDecoder(H, per, max_iters, kernel_variant=variant, **kw)
        err, conv, llr, its = dec.decode_batch_host(syn_bs, want_llr=True, want_iters=True)
        # a second call without LLRs exercises the other kernel instantiation
        err2, conv2, _, its2 = dec.decode_batch_host(syn_bs, want_llr=False, want_iters=True)
        tag = f"[kernel_variant={variant}] "
        assert np.array_equal(conv, oconv), tag + f"converged flags differ at {np.nonzero(conv != oconv)[0][:10]}"
        assert np.array_equal(its, oits), tag + f"iteration counts differ at {np.nonzero(its != oits)[0][:10]}"
        assert np.array_equal(err, oerr), tag + f"hard decisions differ in rows {np.unique(np.nonzero(err != oerr)[0])[:10]}"
        assert np.array_equal(err2, err) and np.array_equal(conv2, conv) and np.array_equal(its2, its), tag
        fin = np.isfinite(ollr)
        assert np.array_equal(np.isfinite(llr), fin), tag
        assert np.array_equal(llr[~fin], ollr[~fin]), tag      # +-Inf must match exactly
        if fin.any():
            assert np.max(np.abs(llr[fin] - ollr[fin])) <= LLR_TOL, tag
        dec.close()
    return err, conv, its


def test_c2_regular_3_6_n1008_batch4096(ldpc, gpu):
    """BASELINE configs[1]: (3,6)-regular n=1008, per=0.01, 50 iters, batch 4096, bit-exact vs CPU."""
    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    e = ldpc.codes.random_errors(1008, 4096, 0.01, seed=11)
    syn = ldpc.codes.syndromes_of(H, e)
    err, conv, its = assert_parity(ldpc, H, 0.01, 50, syn)
    assert conv.mean() > 0.9


def test_reference_test_code_1000_10_9(ldpc, gpu):
    """The code of test/test_bp_decoder.jl: parity_check_matrix(1000,10,9), per=0.01, 100 iters."""
    H = ldpc.codes.parity_check_csc(1000, 10, 9)
    e = ldpc.codes.random_errors(1000, 300, 0.01, seed=12)
    syn = ldpc.codes.syndromes_of(H, e)
    err, conv, its = assert_parity(ldpc, H, 0.01, 100, syn)
    assert conv.all() and np.array_equal(err, e)            # test_bp_decoder.jl:46-51


@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 129])
def test_ragged_batches(ldpc, gpu, B):
    H = ldpc.codes.parity_check_csc(504, 6, 3)
    e = ldpc.codes.random_errors(504, B, 0.03, seed=100 + B)
    assert_parity(ldpc, H, 0.03, 30, ldpc.codes.syndromes_of(H, e))


@pytest.mark.parametrize("per", [0.06, 0.10, 0.2])
def test_non_convergent_full_iterations(ldpc, gpu, per):
    """Above threshold nothing converges: all max_iters run, messages reach Inf/NaN territory."""
    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    e = ldpc.codes.random_errors(1008, 200, per, seed=int(per * 1000))
    assert_parity(ldpc, H, per, 50, ldpc.codes.syndromes_of(H, e))


@pytest.mark.parametrize("per", [0.0, 1e-12, 0.5, 0.999, 1.0])
def test_extreme_channel_probabilities(ldpc, gpu, per):
    """per = 0 / 1 give odds 0 / Inf; the NaN-reset (belief_propagation.jl:158-160,174-176) must match."""
    rng = np.random.default_rng(5)
    H = ldpc.codes.parity_check_csc(96, 6, 3)
    syn = rng.integers(0, 2, (70, 48)).astype(np.uint8)
    assert_parity(ldpc, H, per, 12, syn)


@pytest.mark.parametrize("wpt", [4, 8, 16])
def test_waves_per_tile_variants(ldpc, gpu, wpt):
    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    e = ldpc.codes.random_errors(1008, 200, 0.04, seed=77)
    assert_parity(ldpc, H, 0.04, 40, ldpc.codes.syndromes_of(H, e), waves_per_tile=wpt)


def test_irregular_graphs_with_empty_and_heavy_nodes(ldpc, gpu):
    """Degree-0 checks/bits, a degree-40 check (> widest register bucket) and a degree-20 bit."""
    rng = np.random.default_rng(9)
    s, n = 60, 120
    H = (rng.random((s, n)) < 0.05).astype(np.uint8)
    H[3, :] = 0
    H[:, 7] = 0
    H[5, :40] = 1          # check of degree >= 40 -> O(deg^2) path
    H[10:30, 50] = 1       # bit of degree >= 20  -> O(deg^2) path
    e = (rng.random((150, n)) < 0.03).astype(np.uint8)
    syn = (e.astype(np.int64) @ H.T.astype(np.int64) % 2).astype(np.uint8)
    syn[100:] = rng.integers(0, 2, (50, s))
    assert_parity(ldpc, H, 0.03, 25, syn)


def test_mid_degree_buckets(ldpc, gpu):
    """Degrees 9..16 and 17..32 select the wider register buckets."""
    for (n, wr, wc, per) in [(480, 12, 6, 0.01), (960, 24, 5, 0.004)]:
        H = ldpc.codes.parity_check_csc(n, wr, wc)
        e = ldpc.codes.random_errors(n, 130, per, seed=n)
        assert_parity(ldpc, H, per, 30, ldpc.codes.syndromes_of(H, e))


def test_bb72_code(ldpc, gpu):
    """BASELINE configs[4] BP stage: [[72,12,6]] bivariate bicycle H_X, per=0.005."""
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    e = ldpc.codes.random_errors(72, 5000, 0.005, seed=72)
    assert_parity(ldpc, HX, 0.005, 50, ldpc.codes.syndromes_of(HX, e))


def test_non_binary_syndrome_entries(ldpc, gpu):
    """Entries 2/3 keep their parity for the sign and can never converge (:136,:181)."""
    H = ldpc.codes.parity_check_csc(96, 6, 3)
    syn = np.zeros((66, 48), dtype=np.uint8)
    syn[1, 0] = 2
    syn[2, 5] = 3
    syn[65, 47] = 2
    err, conv, its = assert_parity(ldpc, H, 0.02, 9, syn)
    assert conv[0] and not conv[1] and not conv[2] and not conv[65]
    assert its[1] == 9


def test_max_iters_zero(ldpc, gpu):
    H = ldpc.codes.parity_check_csc(96, 6, 3)
    dec = ldpc.BeliefPropagationDecoder(H, 0.02, 0)
    err, conv, llr, its = dec.decode_batch_host(np.zeros((5, 48), dtype=np.uint8), want_llr=True, want_iters=True)
    assert not err.any() and not conv.any() and not llr.any() and not its.any()


def test_c3_code_n16384_small_batch(ldpc, gpu):
    """BASELINE configs[2] code (n=16384, m=8192, row weight 8) at oracle-sized batch:
    realistic (per=0.02) and full-50 (per=0.10) workloads."""
    H = ldpc.codes.parity_check_csc(16384, 8, 4)
    for per, B in [(0.02, 96), (0.10, 70)]:
        e = ldpc.codes.random_errors(16384, B, per, seed=int(per * 100))
        err, conv, its = assert_parity(ldpc, H, per, 50, ldpc.codes.syndromes_of(H, e))
        if per == 0.10:
            assert not conv.any() and (its == 50).all()


@pytest.mark.parametrize("take,team_max,t0,t1,cap", [("8", None, None, None, None), ("8", "1", None, None, None),
                                                      ("256", None, None, None, None), (None, None, None, None, None),
                                                      ("8", "1", "40", "40", None), ("8", None, "40", "30", None),
                                                      ("8", "1", "40", "40", "2"), ("4", "1", "32", "0", None)])
def test_straggler_handoff_levels_and_pass_kinds(ldpc, gpu, take, team_max, t0, t1, cap, monkeypatch):
    """A medium batch on a code beyond the LDS, broad iteration distribution: tiles hand their stragglers -- with
    their message columns -- to packed level 1, whose pass resumes them where they stood and hands ITS stragglers
    on to level 2.  Which kernel finishes a level is decided on the device from its count: the node-parallel
    kernel up to LDPC_NODE_TAKE_MAX syndromes, teams of workgroups on the packed tiles above that, one workgroup
    per packed tile where teams are off (LDPC_TEAM_MAX=1) or the stragglers too many.  (take 8: teams / packed
    tiles; 256 and default: node kernel.)  t0 / t1: hand-off thresholds of fresh / level-1 tiles (40: most
    syndromes travel through both levels; t1 = 0: a single level); cap: packed tiles per level, so small that
    levels fill up and tiles must carry on by themselves.  Every syndrome against the oracle, LLRs and
    iteration counts included: a resumed syndrome must come out exactly as if nobody had touched it."""
    for k, v in (("LDPC_NODE_TAKE_MAX", take), ("LDPC_TEAM_MAX", team_max), ("LDPC_DEFER_T0", t0), ("LDPC_DEFER_T1", t1),
                 ("LDPC_DEFER_CAP_TILES", cap)):
        if v is not None:
            monkeypatch.setenv(k, v)
    monkeypatch.setenv("LDPC_NODE_MSG_LDS", "0")         # (by default this code's messages live in LDS and no batch
                                                         # of it ever reaches the tile / team kernels)
    n = 4096
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    B = 2600 if take is None else 1200
    e = ldpc.codes.random_errors(n, B, 0.065, seed=41)
    syn = ldpc.codes.syndromes_of(H, e)
    err, conv, its = assert_parity(ldpc, H, 0.065, 40, syn, kernel_variant=0)
    assert 0.2 < conv.mean() and len(np.unique(its)) > 8   # the hand-off has something to do
    # the same through the forced tile kernel (fresh tiles by one workgroup each, whatever the batch size)
    assert_parity(ldpc, H, 0.065, 40, syn[:700], kernel_variant=1)


@pytest.mark.parametrize("B,per,scatter,cache_mib", [(640, 0.02, False, None), (4160, 0.065, False, None), (3000, 0.10, False, None),
                                                     (640, 0.10, True, None), (2900, 0.065, True, None),
                                                     (4160, 0.065, False, 64), (3000, 0.10, False, 64), (2900, 0.02, True, 64)])
def test_team_kernel_is_bit_identical_to_the_tile_kernel(ldpc, gpu, B, per, scatter, cache_mib, monkeypatch):
    """Several workgroups per tile (agent-scope release / acquire between the sweeps) against one
    workgroup per tile: same node updates, so EVERYTHING must come out bit for bit the same, LLRs
    included -- one stale message row anywhere would show.  n = 4096 beyond the LDS, 10 ... 65 tiles
    of very different iteration counts (uneven load), teams of 8 or 7 workgroups (n = 16384 teams of 32 run
    in test_c3_code_n16384_small_batch).  scatter: the members of a team are dealt over ALL XCDs
    (LDPC_TEAM_SCATTER), so the teams find themselves on several XCDs and every barrier writes the L2
    back -- the path taken if the round-robin placement of workgroups ever changes.  cache_mib: the budget for
    message slots in flight (LDPC_TEAM_CACHE_MIB; 64 MiB = 8 slots of this code): 8 persistent teams that take 6 ... 9
    tiles each from the queue, one after the other in the same slot, instead of the default 32 teams."""
    import torch

    if scatter:
        monkeypatch.setenv("LDPC_TEAM_SCATTER", "1")
    if cache_mib is not None:
        monkeypatch.setenv("LDPC_TEAM_CACHE_MIB", str(cache_mib))
    n = 4096
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    syn = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, per, seed=B))).cuda()
    res = {}
    for variant in (1, 4):
        dec = ldpc.BeliefPropagationDecoder(H, per, 40, kernel_variant=variant)
        for rep in range(3 if variant == 4 else 1):     # the team kernel thrice: placement and timing vary
            err = torch.empty((B, n), dtype=torch.uint8, device="cuda")
            conv = torch.empty(B, dtype=torch.uint8, device="cuda")
            llr = torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda")
            its = torch.empty(B, dtype=torch.int32, device="cuda")
            dec.decode_batch_device(syn, err, conv, llr, its)
            torch.cuda.synchronize()
            out = (err, conv, its, llr.view(torch.int64))
            if variant in res:
                assert all(torch.equal(a, b) for a, b in zip(out, res[variant])), "team kernel not reproducible"
            res[variant] = out
        if variant == 4:
            info = dec.info()
            assert info.last_kernel == 4 and info.last_team_size >= 3     # really several workgroups per tile
            # the grid is whole teams; with more tiles than teams the teams are persistent (a team takes tile after tile)
            assert info.resident_tiles % info.last_team_size == 0
            assert info.resident_tiles >= info.last_team_size * min((B + 63) // 64, 6)
            if cache_mib is not None:
                assert info.resident_tiles == 8 * info.last_team_size     # 8 teams for 45 ... 65 tiles
        else:
            assert dec.info().last_kernel == 1
        dec.close()
    names = ("hard decisions", "converged", "iterations", "LLR bits")
    for a, b, nm in zip(res[1], res[4], names):
        assert torch.equal(a, b), f"{nm} differ between the tile and the team kernel"


@pytest.mark.parametrize("per", [0.03, 0.065, 0.10])
def test_team_rows_in_lds_change_nothing(ldpc, gpu, per, monkeypatch):
    """(4,8)-regular graphs: the members of a persistent team keep the message rows that only they touch in LDS (a bit is
    dealt to a member that owns one of its checks; bp_team_kernels.hpp TeamRows).  With the rows in LDS (default), with
    with rows in the waves' registers as well (up to 32 or 7 a wave; bp_team_kernels.hpp "Rows in REGISTERS") or not, with
    the on-chip rows gathered into whole checks (default), spread as in round 2, or whole checks only; the upper waves
    walking forwards or backwards; nodes loaded singly, in pairs or four bits at a time; with
    every row in the slot (LDPC_TEAM_ROWS=0) and through the tile kernel: the same bits, LLRs included -- at an error
    rate where tiles finish early, one where they hand stragglers on (the rows in LDS are written back for that) and
    one where nothing converges."""
    import torch

    n, B = 4096, 3000
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    syn = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, per, seed=77))).cuda()
    res = []
    # (kernel variant, rows on chip, register rows per wave, how bits are dealt: 1 = to the owner of their first check /
    #  0 = most room / 2 = whole checks only, mirrored order of the upper waves, nodes loaded 2 (1) or 2 and 4 (3) at a time)
    for variant, rows, regs, conc, flip, pairs in ((4, "1", "32", "1", "3", "3"), (4, "1", "0", "1", "3", "3"), (4, "1", "7", "0", "3", "3"),
                                                   (4, "0", "32", "1", "3", "3"), (1, "1", "32", "1", "3", "3"),
                                                   (4, "1", "32", "0", "0", "1"), (4, "1", "32", "2", "1", "3"), (4, "1", "0", "0", "2", "0"),
                                                   (4, "1", "5", "2", "3", "1")):
        monkeypatch.setenv("LDPC_TEAM_ROWS", rows)
        monkeypatch.setenv("LDPC_TEAM_REGS", regs)          # rows in the waves' top registers on top (0 = LDS only)
        monkeypatch.setenv("LDPC_TEAM_CONCENTRATE", conc)
        monkeypatch.setenv("LDPC_TEAM_FLIP", flip)
        monkeypatch.setenv("LDPC_TEAM_PAIRS", pairs)
        dec = ldpc.BeliefPropagationDecoder(H, per, 30, kernel_variant=variant)
        err = torch.empty((B, n), dtype=torch.uint8, device="cuda")
        conv = torch.empty(B, dtype=torch.uint8, device="cuda")
        llr = torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda")
        its = torch.empty(B, dtype=torch.int32, device="cuda")
        dec.decode_batch_device(syn, err, conv, llr, its)
        dec.decode_batch_device(syn, err, conv, None, its)       # ... and the instantiation without LLRs
        torch.cuda.synchronize()
        assert dec.info().last_kernel == variant
        res.append((err, conv, its, llr.view(torch.int64)))
        dec.close()
    for other in res[1:]:
        assert all(torch.equal(a, b) for a, b in zip(res[0], other))


@pytest.mark.parametrize("per,ahead", [(0.04, "1"), (0.04, "32"), (0.065, "1"), (0.065, "20"), (0.10, "64"), (0.10, "0")])
def test_team_running_ahead_changes_nothing(ldpc, gpu, per, ahead, monkeypatch):
    """Two team barriers an iteration instead of three: with at least LDPC_TEAM_AHEAD lanes active a team starts the next
    check sweep while the convergence test is still under way and reads the verdict after that sweep's barrier
    (bp_team_kernels.hpp, TeamParams::ahead_min; the decision words are double-buffered for it).  1 = always (every tile
    that was quiet so far may end with a sweep that was for nothing or hand off an iteration late), 64 = only full tiles, 0 = never: the
    oracle's bits every time -- hard decisions, flags, iteration counts, LLRs -- on a code with rows in LDS (persistent
    teams on a small graph through the cache budget) at error rates where tiles finish early, hand stragglers on, or
    run all iterations."""
    n, B = 4096, 2500
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, per, seed=55))
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=25)
    oerr, oconv, ollr, oits = oc.batchdecode(syn)
    monkeypatch.setenv("LDPC_TEAM_AHEAD", ahead)
    monkeypatch.setenv("LDPC_TEAM_CACHE_KIB", "40000")     # 8 MiB a slot: four persistent teams for the 40 tiles
    monkeypatch.setenv("LDPC_TEAM_MIN_ROWS", "512")
    dec = ldpc.BeliefPropagationDecoder(H, per, 25, kernel_variant=4)
    err, conv, llr, its = dec.decode_batch_host(syn, want_llr=True, want_iters=True)
    assert dec.info().last_kernel == 4
    dec.close()
    assert np.array_equal(err, oerr) and np.array_equal(conv, oconv) and np.array_equal(its, oits)
    fin = np.isfinite(ollr)
    assert np.array_equal(llr[~fin], ollr[~fin]) and np.max(np.abs(llr[fin] - ollr[fin])) <= 1e-5


def test_auto_dispatch_picks_the_kernel_by_code_and_batch(ldpc, gpu, monkeypatch):
    """kernel_variant 0: LDS-resident kernel for a code that fits the LDS; node-parallel kernel with the messages
    in LDS at every batch size for a code whose messages alone fit it (n = 4096); beyond that the node-parallel
    kernel below one tile, the team kernel from there on -- persistent teams whose message slots stay in the
    Infinity Cache, as one team per XCD or, for larger graphs, as a few WIDE teams over all XCDs (n = 32768: four,
    n = 65536: two) -- and the tile kernel beyond n = 131072
    (ldpc_bp_info.last_kernel, numbered like kernel_variant).  Results against the oracle on a sample."""
    small = ldpc.codes.parity_check_csc(1008, 6, 3)
    big = ldpc.codes.parity_check_csc(4096, 8, 4)
    d_mid = ldpc.BeliefPropagationDecoder(big, 0.03, 30)
    d_mid.decode_batch_host(ldpc.codes.syndromes_of(big, ldpc.codes.random_errors(4096, 20000, 0.03, seed=20000)))
    assert d_mid.info().last_kernel == 3
    d_mid.close()
    monkeypatch.setenv("LDPC_NODE_MSG_LDS", "0")    # from here on: as if the messages did not fit the LDS
    d_small = ldpc.BeliefPropagationDecoder(small, 0.01, 50)
    d_small.decode_batch_host(ldpc.codes.syndromes_of(small, ldpc.codes.random_errors(1008, 3000, 0.01, seed=1)))
    assert d_small.info().last_kernel == 2
    d_big = ldpc.BeliefPropagationDecoder(big, 0.03, 30)
    oc = BPOracle(csc=(big.indptr, big.indices), shape=big.shape, per=0.03, max_iters=30)
    for B, want in [(1, 3), (40, 3), (2048, 4), (20000, 4)]:
        syn = ldpc.codes.syndromes_of(big, ldpc.codes.random_errors(4096, B, 0.03, seed=B))
        err, conv, _, its = d_big.decode_batch_host(syn, want_iters=True)
        assert d_big.info().last_kernel == want, (B, d_big.info().last_kernel)
        k = min(B, 200)
        oerr, oconv, _, oits = oc.batchdecode(syn[:k])
        assert np.array_equal(err[:k], oerr) and np.array_equal(conv[:k], oconv) and np.array_equal(its[:k], oits)
    d_big.close()
    # n = 32768: eight message slots are 512 MiB, twice the Infinity Cache; with a quarter of the rows on chip FOUR slots
    # fit it: four WIDE teams of 64 workgroups, each dealt over all XCDs (round 4), 19 tiles for the 4 teams -- and, with
    # LLRs, hard decisions and LLRs against the oracle (the wide teams' variable sweep captures the odds in position order)
    huge = ldpc.codes.parity_check_csc(32768, 8, 4)
    d_huge = ldpc.BeliefPropagationDecoder(huge, 0.03, 30)
    syn = ldpc.codes.syndromes_of(huge, ldpc.codes.random_errors(32768, 1200, 0.03, seed=5))
    err, conv, _, its = d_huge.decode_batch_host(syn, want_iters=True)
    info = d_huge.info()
    assert info.last_kernel == 4 and info.last_team_size == 64 and info.resident_tiles == 4 * 64, (info.last_kernel, info.last_team_size, info.resident_tiles)
    assert info.last_rows_on_chip >= 0.9 * huge.nnz // 4
    oc = BPOracle(csc=(huge.indptr, huge.indices), shape=huge.shape, per=0.03, max_iters=30)
    oerr, oconv, ollr, oits = oc.batchdecode(syn[:100])
    assert np.array_equal(err[:100], oerr) and np.array_equal(conv[:100], oconv) and np.array_equal(its[:100], oits)
    err2, conv2, llr2, its2 = d_huge.decode_batch_host(syn, want_llr=True, want_iters=True)
    assert np.array_equal(err2, err) and np.array_equal(conv2, conv) and np.array_equal(its2, its)
    fin = np.isfinite(ollr)
    assert np.array_equal(llr2[:100][~fin], ollr[~fin]) and np.max(np.abs(llr2[:100][fin] - ollr[fin])) <= 1e-6
    d_huge.close()
    # n = 65536 (128 MiB a slot): two wide teams of 128, against the tile kernel bit for bit and the oracle on a sample
    vast = ldpc.codes.parity_check_csc(65536, 8, 4)
    syn = ldpc.codes.syndromes_of(vast, ldpc.codes.random_errors(65536, 700, 0.03, seed=6))
    res = {}
    for variant in (0, 1):
        d_vast = ldpc.BeliefPropagationDecoder(vast, 0.03, 20, kernel_variant=variant)
        res[variant] = d_vast.decode_batch_host(syn, want_llr=True, want_iters=True)
        info = d_vast.info()
        assert (info.last_kernel, info.last_team_size) == ((4, 128) if variant == 0 else (1, 1)), (variant, info.last_kernel, info.last_team_size)
        d_vast.close()
    assert all(np.array_equal(a, b) for a, b in zip(res[0][:2] + res[0][3:], res[1][:2] + res[1][3:]))
    assert np.array_equal(res[0][2].view(np.int64), res[1][2].view(np.int64))
    oc = BPOracle(csc=(vast.indptr, vast.indices), shape=vast.shape, per=0.03, max_iters=20)
    oerr, oconv, _, oits = oc.batchdecode(syn[:40], want_llr=False)
    assert np.array_equal(res[0][0][:40], oerr) and np.array_equal(res[0][1][:40], oconv) and np.array_equal(res[0][3][:40], oits)


def test_device_resident_entry(ldpc, gpu):
    """ldpc_bp_decode_batch_device with HBM-resident tensors on a non-default stream."""
    import torch

    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    e = ldpc.codes.random_errors(1008, 700, 0.02, seed=3)
    syn = ldpc.codes.syndromes_of(H, e)
    dec = ldpc.BeliefPropagationDecoder(H, 0.02, 50)
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        d_syn = torch.from_numpy(syn).to(dev)
        d_err = torch.empty((700, 1008), dtype=torch.uint8, device=dev)
        d_conv = torch.empty(700, dtype=torch.uint8, device=dev)
        d_llr = torch.empty((700, 1008), dtype=torch.float64, device=dev)
        d_it = torch.empty(700, dtype=torch.int32, device=dev)
        dec.decode_batch_device(d_syn, d_err, d_conv, d_llr, d_it)
    st.synchronize()
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.02, max_iters=50)
    oerr, oconv, ollr, oits = oc.batchdecode(syn)
    assert np.array_equal(d_err.cpu().numpy(), oerr)
    assert np.array_equal(d_conv.cpu().numpy(), oconv)
    assert np.array_equal(d_it.cpu().numpy(), oits)
    llr = d_llr.cpu().numpy()
    fin = np.isfinite(ollr)
    assert np.array_equal(llr[~fin], ollr[~fin]) and np.max(np.abs(llr[fin] - ollr[fin])) <= LLR_TOL
    sweep_ms, total_ms, sum_iters = dec.last_timing()
    assert sum_iters == int(oits.sum()) and 0 < sweep_ms <= total_ms


def test_degenerate_shapes(ldpc, gpu):
    """No checks at all, no edges at all, a single bit: the loops of decode! simply do not run."""
    rng = np.random.default_rng(11)
    for shape, dens in [((0, 5), 0.0), ((4, 6), 0.0), ((1, 1), 1.0), ((3, 1), 1.0)]:
        H = sp.csc_matrix((rng.random(shape) < dens).astype(np.uint8)) if dens < 1 else sp.csc_matrix(np.ones(shape, dtype=np.uint8))
        syn = rng.integers(0, 2, (70, shape[0])).astype(np.uint8)
        for per in (0.1, 0.7):
            assert_parity(ldpc, H, per, 5, syn)


def test_lost_team_is_reported_or_repaired_never_silent(ldpc, gpu, monkeypatch):
    """A team barrier that times out raises a fault word (here injected): the synchronous host entry decodes
    once more without teams and returns correct results; the asynchronous device entry reports the fault at
    the next call on the handle."""
    import torch

    n = 4096
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, 700, 0.03, seed=9))
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.03, max_iters=30)
    oerr, oconv, _, oits = oc.batchdecode(syn)
    monkeypatch.setenv("LDPC_TEAM_INJECT_FAULT", "1")
    dec = ldpc.BeliefPropagationDecoder(H, 0.03, 30, kernel_variant=4)
    err, conv, _, its = dec.decode_batch_host(syn, want_iters=True)         # repaired inside the call
    assert np.array_equal(err, oerr) and np.array_equal(conv, oconv) and np.array_equal(its, oits)
    assert dec.info().last_kernel == 1                                       # teams are off for this decoder now
    dec.close()
    dec = ldpc.BeliefPropagationDecoder(H, 0.03, 30, kernel_variant=4)
    d_syn = torch.from_numpy(syn).cuda()
    d_err = torch.empty((700, n), dtype=torch.uint8, device="cuda")
    d_conv = torch.empty(700, dtype=torch.uint8, device="cuda")
    dec.decode_batch_device(d_syn, d_err, d_conv)                            # asynchronous: cannot know yet
    torch.cuda.synchronize()
    with pytest.raises(ldpc.LdpcError, match="call #1 on this decoder lost a workgroup"):
        dec.decode_batch_device(d_syn, d_err, d_conv)                        # refused: enqueues nothing
    dec.decode_batch_device(d_syn, d_err, d_conv)                            # reported once; teams are off now
    torch.cuda.synchronize()
    assert dec.info().last_kernel == 1
    assert np.array_equal(d_err.cpu().numpy(), oerr) and np.array_equal(d_conv.cpu().numpy(), oconv)
    dec.close()
    # ldpc_bp_last_status: the asynchronous caller asks instead of waiting for the next call to tell
    dec = ldpc.BeliefPropagationDecoder(H, 0.03, 30, kernel_variant=4)
    dec.last_status()                                                        # nothing enqueued yet: fine
    d_err.fill_(7)
    dec.decode_batch_device(d_syn, d_err, d_conv)
    with pytest.raises(ldpc.LdpcError, match=r"call #1 on this decoder lost a workgroup.*up to call #1"):
        dec.last_status()                                                    # (synchronises the handle itself)
    dec.last_status()                                                        # reported exactly once
    dec.decode_batch_device(d_syn, d_err, d_conv)
    dec.last_status()
    assert np.array_equal(d_err.cpu().numpy(), oerr) and np.array_equal(d_conv.cpu().numpy(), oconv)
    dec.close()
    monkeypatch.delenv("LDPC_TEAM_INJECT_FAULT")


def test_team_that_is_incomplete_at_launch_costs_milliseconds_not_seconds(ldpc, gpu, monkeypatch):
    """Launch-time roll call (bp_team_kernels.hpp team_rollcall): a member that never gets its CU -- injected: the last
    member of team 0 stays away -- makes the team give up within the roll call's bound (20 ms), before anything has
    been read or written; the synchronous host entry then decodes the batch with the tile kernel at once, where a
    timed-out team barrier would have cost 10 s.  The asynchronous entry reports it like any other team fault."""
    import time

    import torch

    n = 16384
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, 640, 0.04, seed=19))
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.04, max_iters=12)
    oerr, oconv, _, oits = oc.batchdecode(syn, want_llr=False)
    monkeypatch.setenv("LDPC_TEAM_INJECT_FAULT", "2")
    dec = ldpc.BeliefPropagationDecoder(H, 0.04, 12, kernel_variant=4)
    dec.decode_batch_host(syn[:64])                                          # (first call: allocations, module load)
    dec.close()
    dec = ldpc.BeliefPropagationDecoder(H, 0.04, 12, kernel_variant=4)
    t0 = time.perf_counter()
    err, conv, _, its = dec.decode_batch_host(syn, want_iters=True)          # team grid, roll call fails, tile kernel
    dt = time.perf_counter() - t0
    assert np.array_equal(err, oerr) and np.array_equal(conv, oconv) and np.array_equal(its, oits)
    assert dec.info().last_kernel == 1                                       # teams are off for this decoder now
    assert dt < 3.0, f"the fallback took {dt:.2f} s: the roll call did not bound the wait"
    dec.close()
    dec = ldpc.BeliefPropagationDecoder(H, 0.04, 12, kernel_variant=4)
    d_syn = torch.from_numpy(syn).cuda()
    d_err = torch.full((640, n), 7, dtype=torch.uint8, device="cuda")
    d_conv = torch.full((640,), 7, dtype=torch.uint8, device="cuda")
    dec.decode_batch_device(d_syn, d_err, d_conv)
    t0 = time.perf_counter()
    with pytest.raises(ldpc.LdpcError, match=r"call #1 on this decoder found a team of workgroups incomplete at launch"):
        dec.last_status()
    assert time.perf_counter() - t0 < 3.0
    assert bool((d_conv[:64] == 7).all())          # the incomplete team (it starts on tile 0) wrote nothing at all
    dec.decode_batch_device(d_syn, d_err, d_conv)                            # teams are off: the tile kernel
    dec.last_status()
    assert np.array_equal(d_err.cpu().numpy(), oerr) and np.array_equal(d_conv.cpu().numpy(), oconv)
    dec.close()
    monkeypatch.delenv("LDPC_TEAM_INJECT_FAULT")


@pytest.mark.parametrize("n,wr,wc", [(16000, 10, 5), (24000, 24, 3)])
def test_wide_degree_buckets_on_large_codes_small_batches(ldpc, gpu, n, wr, wc):
    """Codes beyond the LDS whose nodes need the wide register buckets (check degree 10 / bit degree 5: the 16-wide
    instantiations, 131-167 VGPRs; check degree 24: the 32-wide one, 256 VGPRs -- ONE 8-wave workgroup per CU), nnz >
    64k, at the batch sizes where the cost model picks teams: a team grid must never be larger than what those
    instantiations keep resident (round 1 launched 8 x team workgroups for <= 4 tiles and the cooperative launch
    refused it: a plain decode! failed with LDPC_ERR_HIP).  Batch 1, 64 and 256 against the oracle."""
    H = ldpc.codes.parity_check_csc(n, wr, wc)
    assert H.nnz > 64000
    E = ldpc.codes.random_errors(n, 256, 0.02, seed=n)
    syn = ldpc.codes.syndromes_of(H, E)
    for B in (1, 64, 256):
        for variant in (0, 4):
            assert_parity(ldpc, H, 0.02, 12, syn[:B], kernel_variant=variant)
    dec = ldpc.BeliefPropagationDecoder(H, 0.02, 12, kernel_variant=4)
    dec.decode_batch_host(syn[:64])
    info = dec.info()
    assert info.last_kernel == 4 and info.last_team_size >= 3 and info.resident_tiles <= 256   # really a team grid, and resident
    dec.close()


def test_one_handle_driven_from_two_streams_alternately(ldpc, gpu):
    """Calls on one handle execute in call order whatever streams they are given (include/ldpc_mi355x.h): two
    non-blocking torch streams alternate on ONE decoder without any host synchronisation in between -- the
    single-kernel paths rely on the kernel of call N having zeroed the control slot of call N+1, and every path
    shares the handle's workspace.  LDS kernel (n = 1008), node kernel (n = 16384, 40 syndromes) and team / tile
    kernel (n = 16384, 700 syndromes): every call's outputs must equal the reference decode."""
    import torch

    for n, wr, wc, B, per in [(1008, 6, 3, 3000, 0.02), (16384, 8, 4, 40, 0.03), (16384, 8, 4, 700, 0.03)]:
        H = ldpc.codes.parity_check_csc(n, wr, wc)
        S = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, 2 * B, per, seed=B))).cuda()
        dec = ldpc.BeliefPropagationDecoder(H, per, 30)
        refs = []
        for half in range(2):
            e = torch.empty((B, n), dtype=torch.uint8, device="cuda")
            c = torch.empty(B, dtype=torch.uint8, device="cuda")
            i = torch.empty(B, dtype=torch.int32, device="cuda")
            dec.decode_batch_device(S[half * B:(half + 1) * B], e, c, None, i)
            torch.cuda.synchronize()
            refs.append((e, c, i))
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = []
        torch.cuda.synchronize()
        for k in range(24):
            half = k % 2 if k % 3 else (k // 3) % 2          # an irregular alternation of inputs ...
            st = streams[k % 2]                               # ... and a strict alternation of streams
            e = torch.full((B, n), 9, dtype=torch.uint8, device="cuda")
            c = torch.full((B,), 9, dtype=torch.uint8, device="cuda")
            i = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            st.wait_stream(torch.cuda.current_stream())       # (the fills above ran on the current stream)
            dec.decode_batch_device(S[half * B:(half + 1) * B], e, c, None, i, stream=st.cuda_stream)
            outs.append((half, e, c, i))
        dec.last_status()                                     # waits for the last call, hence for all of them
        torch.cuda.synchronize()
        for k, (half, e, c, i) in enumerate(outs):
            assert torch.equal(e, refs[half][0]) and torch.equal(c, refs[half][1]) and torch.equal(i, refs[half][2]), (n, B, k)
        dec.close()


@pytest.mark.parametrize("n,shift", [(1008, 0), (1008, 1), (1008, 2), (222, 0), (222, 3)])
def test_device_arrays_of_any_alignment(ldpc, gpu, n, shift):
    """The pack / unpack kernels of the tile and team kernels take four checks (bits) per lane where the caller's device
    arrays allow 4-byte accesses (s, n multiples of 4 and 4-byte aligned pointers) and one per lane otherwise
    (ldpc_mi355x.hip syn_v4 / err_v4): syndromes and decisions at odd addresses and an n that is no multiple of 4, a ragged
    batch, through the HBM-streaming and the team kernel, against the oracle.  (batchdecode! takes any matrix:
    belief_propagation.jl:220-231.)"""
    import torch

    H = ldpc.codes.parity_check_csc(n, 6, 3)
    s_ = H.shape[0]
    B, per, iters = 200, 0.03, 20
    syn_h = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, per, seed=n + shift))
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=iters)
    oerr, oconv, ollr, oits = oc.batchdecode(syn_h, want_llr=True)
    sbuf = torch.zeros(B * s_ + 8, dtype=torch.uint8, device="cuda"); ebuf = torch.full((B * n + 8,), 7, dtype=torch.uint8, device="cuda")
    syn = sbuf[shift:shift + B * s_].view(B, s_); err = ebuf[shift:shift + B * n].view(B, n)
    syn.copy_(torch.from_numpy(syn_h).cuda())
    assert syn.data_ptr() % 4 == shift % 4 and err.data_ptr() % 4 == shift % 4
    for variant in (1, 4):
        dec = ldpc.BeliefPropagationDecoder(H, per, iters, kernel_variant=variant)
        conv = torch.empty(B, dtype=torch.uint8, device="cuda"); its = torch.empty(B, dtype=torch.int32, device="cuda")
        llr = torch.empty((B, n), dtype=torch.float64, device="cuda")
        ebuf.fill_(7)
        dec.decode_batch_device(syn, err, conv, llr, its)
        dec.last_status()
        assert np.array_equal(err.cpu().numpy(), oerr) and np.array_equal(conv.cpu().numpy(), oconv) and np.array_equal(its.cpu().numpy(), oits), variant
        assert bool((ebuf[:shift] == 7).all()) and bool((ebuf[shift + B * n:] == 7).all()), "bytes outside the caller's array written"
        g = llr.cpu().numpy(); fin = np.isfinite(ollr)
        assert np.array_equal(g[~fin], ollr[~fin]) and np.max(np.abs(g[fin] - ollr[fin])) <= LLR_TOL
        dec.close()


def test_llrs_from_the_cut_odds_without_the_library_log(ldpc, gpu):
    """bp_kernels.hpp llr_cut: log_probabs[j] = log(1 / T) (belief_propagation.jl:163) from the posterior odds cut to 21
    significant bits, by frexp, one division and seven terms of 2 atanh instead of a division and the library's log (the
    decoder that turns the team kernel's captured odds into LLRs was bound by those).  On the device, over everything a
    double can be: against the library's log(1 / .) of the same cut odds to 1e-12 -- end cases (+Inf wherever 1 / T
    overflows, -Inf, NaN) identical --, and against log(1 / T) of the odds themselves to the 5e-7 the headers promise (2e-6 for denormal odds)."""
    rng = np.random.default_rng(163)
    L = ldpc._capi.lib()
    bits = rng.integers(0, 0x7FF0_0000_0000_0000, size=3_000_000, dtype=np.uint64)          # every exponent, denormals too
    T = np.concatenate([bits.view(np.float64), np.exp(rng.normal(0, 4, 1_000_000)), 1.0 + rng.normal(0, 1e-4, 200_000),
                        2.0 ** np.arange(-1074, 1024, dtype=np.float64), np.nextafter(2.0 ** np.arange(-1022, 1024, dtype=np.float64), 0),
                        [0.0, 5e-324, 2.0 ** -1024, np.nextafter(2.0 ** -1024, 0), np.nextafter(2.0 ** -1024, 1), 2.0 ** -1023, 1.0,
                         np.sqrt(0.5), np.sqrt(2.0), np.nextafter(1.0, 0), np.nextafter(1.0, 2), 1.7976931348623157e308, np.inf, np.nan]])
    T = np.ascontiguousarray(T, dtype=np.float64)
    fast = np.empty_like(T); lib = np.empty_like(T)
    ldpc._capi.check(L.ldpc_debug_llr_check(T.size, T.ctypes.data, fast.ctypes.data, lib.ctypes.data), L)
    fin = np.isfinite(lib)
    assert np.array_equal(np.isnan(fast), np.isnan(lib)) and np.array_equal(fast[~fin & ~np.isnan(lib)], lib[~fin & ~np.isnan(lib)])
    assert np.all(np.isfinite(fast[fin]))
    worst = float(np.max(np.abs(fast[fin] - lib[fin])))
    assert worst <= 1e-12, worst
    with np.errstate(all="ignore"):
        ref = np.log(1.0 / T)                         # the reference's own expression on the uncut odds
    # (the one place where the cut shows in the end cases: 1 / T overflows for T <= 2^-1024 (1 + 2^-53), the cut puts half a
    #  step of 2^-20 back on, so the odds of the lowest step from 2^-1024 on get a finite LLR of 709.78 -- outside this check)
    band = (T >= 2.0 ** -1024) & (T < 2.0 ** -1024 * (1 + 2.0 ** -20))
    rfin = np.isfinite(ref)
    ends = ~rfin & ~np.isnan(ref) & ~band
    assert np.array_equal(rfin[~band], fin[~band]) and np.array_equal(fast[ends], ref[ends])
    rfin &= ~band
    norm = rfin & (T >= 2.0 ** -1022)
    worst_ref = float(np.max(np.abs(fast[norm] - ref[norm])))
    assert worst_ref <= 5e-7, worst_ref
    # (denormal odds with a finite LLR, 2^-1024 <= T < 2^-1022, LLR 708.4 ... 709.8: the upper 32 bits hold 18-19 of their
    #  significant bits, not 21)
    den = rfin & ~norm
    assert den.any() and float(np.max(np.abs(fast[den] - ref[den]))) <= 2e-6


def test_short_division_equals_the_ieee_division_where_the_kernels_take_it(ldpc, gpu):
    """bp_kernels.hpp div_core (LDPC_FAST_DIV): hipcc's double division without v_div_scale / v_div_fixup, taken by
    the check sweep for 2 / (1 + m) with 1 <= 1 + m < 2^500 and for (1 - t) / (1 + t) with |t| < 1.  On the device,
    against `/`, bit for bit: dense random samples of both ranges, their edges (1, the last double below 2^500, t one
    ulp inside +-1, t tiny), and the exact operand pairs of a decode (powers of two, values an ulp apart)."""
    import ctypes

    rng = np.random.default_rng(2024)
    L = ldpc._capi.lib()

    def check(num, den):
        num = np.ascontiguousarray(num, dtype=np.float64); den = np.ascontiguousarray(den, dtype=np.float64)
        a = np.empty_like(num); b = np.empty_like(num)
        ldpc._capi.check(L.ldpc_debug_div_check(num.size, num.ctypes.data, den.ctypes.data, a.ctypes.data, b.ctypes.data), L)
        assert np.array_equal(b, num / den), "the device's `/` is the IEEE division"
        bad = np.flatnonzero(a.view(np.int64) != b.view(np.int64))
        assert bad.size == 0, (num[bad[:4]], den[bad[:4]], a[bad[:4]], b[bad[:4]])

    # :140  2 / (1 + m): m = odds over the whole range a decode meets, log-uniform, and near the edges
    m = np.concatenate([np.exp(rng.uniform(np.log(1e-300), np.log(1e150), 2_000_000)), rng.uniform(0, 4, 500_000),
                        [0.0, 5e-324, 1e-320, 2.0 ** -53, 2.0 ** -52, 1.0, 2.0 ** 52, 2.0 ** 499, np.nextafter(2.0 ** 500, 0) - 1.0]])
    d = 1.0 + m
    d = d[(d >= 1.0) & (d < 2.0 ** 500)]
    check(np.full(d.size, 2.0), d)
    # exponent edges of the denominator: every power of two below 2^500 and its neighbours
    p2 = 2.0 ** np.arange(0, 500)
    dd = np.concatenate([p2, np.nextafter(p2, np.inf), np.nextafter(p2[1:], 0)])
    check(np.full(dd.size, 2.0), dd)
    # :147  (1 - t) / (1 + t), |t| < 1: uniform, concentrated at +-1 and at 0, and the last doubles inside the range
    u = rng.uniform(-1, 1, 2_000_000)
    near1 = 1.0 - np.exp(rng.uniform(np.log(2.0 ** -53), 0, 500_000))
    tiny = np.exp(rng.uniform(np.log(1e-300), 0, 200_000))
    t = np.concatenate([u, near1, -near1, tiny, -tiny, [0.0, -0.0, np.nextafter(1.0, 0), -np.nextafter(1.0, 0), 5e-324, 0.5, -0.5]])
    t = t[np.abs(t) < 1.0]
    check(1.0 - t, 1.0 + t)


def test_llr_precision_is_a_decoder_option_and_the_same_in_every_kernel(ldpc, gpu):
    """ldpc_bp_options.llr_exact.  Default: every kernel returns log(1 / T~) with the posterior odds cut to their upper
    32 bits -- the SAME bits from the tile, LDS, node and team kernels, within 5e-7 of the oracle's log(1 / T), +-Inf
    exact; llr_exact = 1: log(1 / T) itself (<= 1e-9 from the oracle: two libms), again the same bits from every
    kernel.  Decisions, flags and iteration counts do not depend on it.  A mid-size regular code (rows on chip in the
    team kernel) at an error rate where lanes stop at different iterations, and a saturating one (per 1e-6: odds that
    underflow, LLRs of +-Inf)."""
    H = ldpc.codes.parity_check_csc(4032, 8, 4)
    for per in (0.04, 1e-6):
        syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(4032, 700, max(per, 0.01), seed=17))
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=30)
        oerr, oconv, ollr, oits = oc.batchdecode(syn, want_llr=True)
        fin = np.isfinite(ollr)
        for exact in (False, True):
            seen = None
            for variant, kw in ((1, {}), (3, {}), (4, {}), (0, {})):
                dec = ldpc.BeliefPropagationDecoder(H, per, 30, kernel_variant=variant, llr_exact=exact, experiments=True, **kw)
                err, conv, llr, its = dec.decode_batch_host(syn, want_llr=True, want_iters=True)
                dec.close()
                assert np.array_equal(err, oerr) and np.array_equal(conv, oconv) and np.array_equal(its, oits)
                assert np.array_equal(llr[~fin], ollr[~fin])
                worst = float(np.max(np.abs(llr[fin] - ollr[fin]))) if fin.any() else 0.0
                assert worst <= (1e-9 if exact else 1e-6), (variant, exact, worst)
                if seen is None:
                    seen = llr
                else:
                    assert np.array_equal(seen.view(np.int64), llr.view(np.int64)), (variant, exact)
            if not exact and fin.any():
                assert float(np.max(np.abs(seen[fin] - ollr[fin]))) > 0.0   # (it IS a cut: not the exact logarithm)


def test_the_one_team_of_an_xcd_takes_all_its_cus_by_default(ldpc, gpu):
    """ldpc_mi355x.hip team_geometry(): a persistent team that has an XCD to itself gets all 32 CUs down to 1100 message rows a
    member (kTeamMinRowsOne), not only as many members as keep 2048 rows each -- the plan of (4,8)-regular codes between
    n = 9216 and n = 12288.  Through the DEFAULT path (no LDPC_TEAM_MIN_ROWS: the fuzz tool's tiny-graph setting switches
    this rule off), n = 10240 (1280 rows a member), against the oracle, LLRs included."""
    n = 10240
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, 1000, 0.05, seed=10240))
    dec = ldpc.BeliefPropagationDecoder(H, 0.05, 25)
    err, conv, llr, its = dec.decode_batch_host(syn, want_llr=True, want_iters=True)
    info = dec.info()
    assert info.last_kernel == 4 and info.last_team_size == 32 and info.resident_tiles == 8 * 32, (info.last_kernel, info.last_team_size, info.resident_tiles)
    assert info.last_rows_on_chip >= 0.9 * H.nnz // 4
    dec.close()
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.05, max_iters=25)
    oerr, oconv, ollr, oits = oc.batchdecode(syn[:300])
    assert np.array_equal(err[:300], oerr) and np.array_equal(conv[:300], oconv) and np.array_equal(its[:300], oits)
    fin = np.isfinite(ollr)
    assert np.array_equal(llr[:300][~fin], ollr[~fin]) and np.max(np.abs(llr[:300][fin] - ollr[fin])) <= 1e-6
