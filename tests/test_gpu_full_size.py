"""Parity at BASELINE.json's full sizes (configs[2]: n=16384, m=8192, row weight 8, batch 65536,
50 iterations), where the oracle cannot decode everything in seconds:

* size-independent properties on the WHOLE batch -- a converged syndrome's hard decision must
  reproduce the syndrome (H e = s), its iteration count must be < 50 or exactly the iteration of
  convergence, a non-converged one must have run all 50; results must not depend on the
  waves-per-tile / workspace geometry nor on where a syndrome sits in the batch;
* the oracle on a random subset (SURVEY.md 8d parity gate), decoded on the host cores in threads.
"""
import concurrent.futures as cf

import numpy as np
import pytest
import torch

from oracle import BPOracle

pytestmark = pytest.mark.gpu

N, WR, WC, B, ITERS = 16384, 8, 4, 65536, 50


def _device_syndromes(H, per, seed):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    csr = H.tocsr()
    cols = torch.from_numpy(csr.indices.astype(np.int64)).to(dev)
    syn = torch.empty((B, csr.shape[0]), dtype=torch.uint8, device=dev)
    for b0 in range(0, B, 4096):
        e = (torch.rand((4096, N), generator=g, device=dev) < per).to(torch.uint8)
        syn[b0:b0 + 4096] = e[:, cols].view(4096, csr.shape[0], WR).sum(dim=2, dtype=torch.int32).remainder(2).to(torch.uint8)
    return syn, cols


def _decode(ldpc, H, per, syn, **kw):
    dec = ldpc.BeliefPropagationDecoder(H, per, ITERS, **kw)
    dev = syn.device
    err = torch.empty((syn.shape[0], N), dtype=torch.uint8, device=dev)
    conv = torch.empty(syn.shape[0], dtype=torch.uint8, device=dev)
    its = torch.empty(syn.shape[0], dtype=torch.int32, device=dev)
    dec.decode_batch_device(syn, err, conv, None, its)
    torch.cuda.synchronize()
    dec.close()
    return err, conv, its


def _oracle_subset(H, per, syn_np, threads=12):
    def work(chunk):
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=ITERS)
        return oc.batchdecode(chunk, want_llr=False)

    chunks = np.array_split(syn_np, threads)
    with cf.ThreadPoolExecutor(threads) as ex:   # ctypes releases the GIL inside the C oracle
        res = list(ex.map(work, chunks))
    return (np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res]),
            np.concatenate([r[3] for r in res]))


@pytest.mark.parametrize("per,subset", [(0.02, 4096), (0.06, 4096), (0.10, 4096)])
def test_c3_full_batch(ldpc, gpu, per, subset):
    H = ldpc.codes.parity_check_csc(N, WR, WC)
    syn, cols = _device_syndromes(H, per, seed=int(per * 1000))
    err, conv, its = _decode(ldpc, H, per, syn)
    conv_b = conv.bool()
    # (1) converged  =>  H e = s, for every converged syndrome of the batch
    for b0 in range(0, B, 4096):
        s2 = err[b0:b0 + 4096][:, cols].view(4096, H.shape[0], WR).sum(dim=2, dtype=torch.int32).remainder(2).to(torch.uint8)
        ok = (s2 == syn[b0:b0 + 4096]).all(dim=1)
        assert bool((ok == conv_b[b0:b0 + 4096]).all()), "converged flag must equal 'syndrome reproduced'"
    # (2) iteration counts: non-converged ran all 50, converged at most 50
    assert bool((its[~conv_b] == ITERS).all()) and bool((its >= 1).all()) and bool((its <= ITERS).all())
    if per == 0.10:
        assert not bool(conv_b.any())              # above threshold: the full-50 roofline workload
    else:
        assert conv_b.float().mean().item() > 0.99
    if per == 0.06:                                # the waterfall: thousands of syndromes travel through the packed levels
        assert 7.5 < its.float().mean().item() < 10
    # (3) geometry / position independence: another waves-per-tile + fewer workspace slots, and a
    #     permuted batch, must give identical per-syndrome results
    perm = torch.randperm(B, device=syn.device, generator=torch.Generator(device=syn.device).manual_seed(1))[:8192]
    e2, c2, i2 = _decode(ldpc, H, per, syn[perm].contiguous(), waves_per_tile=16, resident_tiles=100)
    assert torch.equal(e2, err[perm]) and torch.equal(c2, conv[perm]) and torch.equal(i2, its[perm])
    # (3b) path independence on the WHOLE batch: auto (stragglers handed through the packed levels, finished by
    #      the node-parallel kernel where few) against the forced all-tile path WITHOUT any hand-off
    e3, c3, i3 = _decode(ldpc, H, per, syn, kernel_variant=1, defer_threshold=-1)
    assert torch.equal(e3, err) and torch.equal(c3, conv) and torch.equal(i3, its)
    del e3, c3, i3
    # (3c) ... and against the tile kernel WITH its hand-off levels (the default path of this code is the team kernel:
    #      8 persistent teams whose message slots stay in the Infinity Cache)
    e4, c4, i4 = _decode(ldpc, H, per, syn, kernel_variant=1)
    assert torch.equal(e4, err) and torch.equal(c4, conv) and torch.equal(i4, its)
    del e4, c4, i4
    # (4) the oracle on a random subset
    idx = np.sort(np.random.default_rng(7).choice(B, subset, replace=False))
    tidx = torch.from_numpy(idx).to(syn.device)
    oerr, oconv, oits = _oracle_subset(H, per, syn[tidx].cpu().numpy())
    assert np.array_equal(conv[tidx].cpu().numpy(), oconv)
    assert np.array_equal(its[tidx].cpu().numpy(), oits)
    assert np.array_equal(err[tidx].cpu().numpy(), oerr)


@pytest.mark.parametrize("Bp", [(1, 0.06), (130, 0.07), (700, 0.06)])
def test_c3_code_team_kernel_is_bit_identical_to_the_tile_kernel(ldpc, gpu, Bp):
    """The C3 code at batches the team kernel takes -- one syndrome (64 workgroups on the one tile, dealt over
    all XCDs), 3 tiles, 11 tiles (teams of 32 inside one XCD): tile kernel, team kernel and auto-dispatch agree
    bit for bit, LLR bit patterns included (the oracle would need minutes for these)."""
    B, per = Bp
    H = ldpc.codes.parity_check_csc(N, WR, WC)
    syn = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(N, B, per, seed=B))).cuda()
    res = {}
    for variant in (1, 4, 0):
        dec = ldpc.BeliefPropagationDecoder(H, per, ITERS, kernel_variant=variant)
        err = torch.empty((B, N), dtype=torch.uint8, device="cuda")
        conv = torch.empty(B, dtype=torch.uint8, device="cuda")
        llr = torch.full((B, N), float("nan"), dtype=torch.float64, device="cuda")
        its = torch.empty(B, dtype=torch.int32, device="cuda")
        dec.decode_batch_device(syn, err, conv, llr, its)
        torch.cuda.synchronize()
        res[variant] = (err, conv, its, llr.view(torch.int64))
        if variant != 1:
            assert dec.info().last_kernel == 4 and dec.info().last_team_size >= 32
        dec.close()
    for v in (4, 0):
        assert all(torch.equal(a, b) for a, b in zip(res[1], res[v]))


def _syndromes_on_device(H, wr, batch, per, seed):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    csr = H.tocsr()
    csr.sort_indices()
    n = H.shape[1]
    cols = torch.from_numpy(csr.indices.astype(np.int64)).to(dev)
    syn = torch.empty((batch, csr.shape[0]), dtype=torch.uint8, device=dev)
    for b0 in range(0, batch, 4096):
        e = (torch.rand((4096, n), generator=g, device=dev) < per).to(torch.uint8)
        syn[b0:b0 + 4096] = e[:, cols].view(4096, csr.shape[0], wr).sum(dim=2, dtype=torch.int32).remainder(2).to(torch.uint8)
    return syn, cols


@pytest.mark.parametrize("n,wr,wc,batch,per", [(16380, 6, 3, 65536, 0.10), (16380, 6, 3, 65536, 0.055),
                                              (16000, 10, 5, 32768, 0.10), (16000, 10, 5, 32768, 0.05),
                                              # round 4: the other pairs of north_star's range (row weight 6 ... 10): (3,9) and (4,10)
                                              (16380, 9, 3, 32768, 0.10), (16380, 9, 3, 32768, 0.02), (16380, 10, 4, 32768, 0.10)])
def test_other_regular_codes_keep_rows_in_lds_full_batch(ldpc, gpu, n, wr, wc, batch, per):
    """Rows in LDS beyond the (4,8) code: a (3,6)-regular code of the C3 size and bench.py's wide_16000_10_5
    ((5,10)-regular, the 16-wide register bucket) at full batch through the default path -- persistent teams whose
    members keep the message rows only they touch in LDS (instantiations <6,3> and <10,5>).  Whole-batch properties,
    bit-equality with the tile kernel (no teams, no hand-off), and the oracle on 1,024 syndromes; per above the
    threshold (every syndrome runs all 50 iterations) and below it (early exit, stragglers through the levels)."""
    H = ldpc.codes.parity_check_csc(n, wr, wc)
    syn, cols = _syndromes_on_device(H, wr, batch, per, seed=n + int(per * 1000))
    dec = ldpc.BeliefPropagationDecoder(H, per, ITERS)
    err = torch.empty((batch, n), dtype=torch.uint8, device=syn.device)
    conv = torch.empty(batch, dtype=torch.uint8, device=syn.device)
    its = torch.empty(batch, dtype=torch.int32, device=syn.device)
    dec.decode_batch_device(syn, err, conv, None, its)
    dec.last_status()
    info = dec.info()
    assert info.last_kernel == 4 and info.last_team_size >= 16 and info.last_lds_rows >= 200, (info.last_kernel, info.last_team_size, info.last_lds_rows)
    # every bit's first edge (the whole checks of the first block) is on chip, in LDS or in registers, as far as the
    # members' capacity (312 + 8 x 32 rows each) goes
    assert info.last_rows_on_chip >= 0.9 * min(H.nnz // wc, info.last_team_size * 568), (info.last_rows_on_chip, H.nnz, info.last_team_size)
    dec.close()
    conv_b = conv.bool()
    for b0 in range(0, batch, 4096):
        s2 = err[b0:b0 + 4096][:, cols].view(4096, H.shape[0], wr).sum(dim=2, dtype=torch.int32).remainder(2).to(torch.uint8)
        ok = (s2 == syn[b0:b0 + 4096]).all(dim=1)
        assert bool((ok == conv_b[b0:b0 + 4096]).all()), "converged flag must equal 'syndrome reproduced'"
    assert bool((its[~conv_b] == ITERS).all()) and bool((its >= 1).all()) and bool((its <= ITERS).all())
    if per >= 0.10:
        assert conv_b.float().mean().item() < 0.01      # above threshold: (nearly) everything runs all 50 iterations
    else:
        assert conv_b.float().mean().item() > 0.9
    tile = ldpc.BeliefPropagationDecoder(H, per, ITERS, kernel_variant=1, defer_threshold=-1)
    e2 = torch.empty_like(err); c2 = torch.empty_like(conv); i2 = torch.empty_like(its)
    tile.decode_batch_device(syn, e2, c2, None, i2)
    tile.last_status()
    assert tile.info().last_kernel == 1
    tile.close()
    assert torch.equal(e2, err) and torch.equal(c2, conv) and torch.equal(i2, its)
    del e2, c2, i2
    idx = np.sort(np.random.default_rng(n).choice(batch, 1024, replace=False))
    tidx = torch.from_numpy(idx).to(syn.device)

    def work(chunk):
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=ITERS)
        return oc.batchdecode(chunk, want_llr=False)

    with cf.ThreadPoolExecutor(12) as ex:
        res = list(ex.map(work, np.array_split(syn[tidx].cpu().numpy(), 12)))
    oerr, oconv, oits = (np.concatenate([r[k] for r in res]) for k in (0, 1, 3))
    assert np.array_equal(conv[tidx].cpu().numpy(), oconv) and np.array_equal(its[tidx].cpu().numpy(), oits)
    assert np.array_equal(err[tidx].cpu().numpy(), oerr)


@pytest.mark.parametrize("per,exact", [(0.10, False), (0.02, False), (0.06, True)])
def test_c3_full_batch_with_llrs(ldpc, gpu, per, exact):
    """The LLR-producing instantiation at the C3 size (the reference's decode! always fills scratch.log_probabs,
    belief_propagation.jl:163; BP+OSD reads them): the whole batch through the default path -- persistent teams that
    capture the posterior odds of every active lane in every iteration (their upper 32 bits; all 64 with llr_exact),
    four positions of the dealt bit order per store, unpacked through the position map.  Hard decisions, flags and iteration counts must be what the
    call without LLRs gives, bit for bit; the LLRs must be the tile kernel's bit for bit on a slice of the batch (every
    kernel cuts the odds the same way), and the oracle's to 1e-5 (default) / 1e-9 (exact) with +-Inf exact on a sample --
    per 0.10: every lane stops at iteration 50; 0.02: lanes stop at different iterations, tiles hand stragglers on;
    0.06 (exact): the waterfall."""
    H = ldpc.codes.parity_check_csc(N, WR, WC)
    syn, cols = _device_syndromes(H, per, seed=int(per * 1000) + 7)
    err0, conv0, its0 = _decode(ldpc, H, per, syn)
    dec = ldpc.BeliefPropagationDecoder(H, per, ITERS, llr_exact=exact)
    err = torch.empty((B, N), dtype=torch.uint8, device=syn.device)
    conv = torch.empty(B, dtype=torch.uint8, device=syn.device)
    its = torch.empty(B, dtype=torch.int32, device=syn.device)
    llr = torch.full((B, N), float("nan"), dtype=torch.float64, device=syn.device)
    dec.decode_batch_device(syn, err, conv, llr, its)
    dec.last_status()
    assert dec.info().last_kernel == 4 and dec.info().last_team_size == 32
    dec.close()
    assert torch.equal(err, err0) and torch.equal(conv, conv0) and torch.equal(its, its0)
    del err0, conv0, its0
    assert not bool(torch.isnan(llr).any())
    # the sign of an LLR is the hard decision (T >= 1 <=> log(1 / T) <= 0), on the whole batch
    for b0 in range(0, B, 8192):
        assert bool(((llr[b0:b0 + 8192] <= 0) == (err[b0:b0 + 8192] == 1)).all())
    # the tile kernel (no teams, bit-major LLR rows) on a slice from the middle of the batch: the same bits
    sl = slice(30000, 30000 + 2048)
    tile = ldpc.BeliefPropagationDecoder(H, per, ITERS, kernel_variant=1, llr_exact=exact)
    e2 = torch.empty((2048, N), dtype=torch.uint8, device=syn.device); c2 = torch.empty(2048, dtype=torch.uint8, device=syn.device)
    l2 = torch.empty((2048, N), dtype=torch.float64, device=syn.device)
    tile.decode_batch_device(syn[sl].contiguous(), e2, c2, l2, None)
    tile.last_status()
    tile.close()
    assert torch.equal(e2, err[sl]) and torch.equal(l2.view(torch.int64), llr[sl].view(torch.int64))
    del e2, c2, l2
    # the oracle on a sample
    idx = np.sort(np.random.default_rng(11).choice(B, 192, replace=False))
    tidx = torch.from_numpy(idx).to(syn.device)
    h_syn = syn[tidx].cpu().numpy()

    def work(chunk):
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=ITERS)
        return oc.batchdecode(chunk, want_llr=True)

    with cf.ThreadPoolExecutor(12) as ex:
        res = list(ex.map(work, np.array_split(h_syn, 12)))
    oerr, oconv, ollr, oits = (np.concatenate([r[k] for r in res]) for k in range(4))
    assert np.array_equal(err[tidx].cpu().numpy(), oerr) and np.array_equal(conv[tidx].cpu().numpy(), oconv)
    assert np.array_equal(its[tidx].cpu().numpy(), oits)
    g = llr[tidx].cpu().numpy()
    fin = np.isfinite(ollr)
    assert np.array_equal(g[~fin], ollr[~fin])
    worst = float(np.max(np.abs(g[fin] - ollr[fin])))
    assert worst <= (1e-9 if exact else 1e-5), worst
    if not exact:
        assert worst <= 1e-6, worst      # (the cut is 2^-21 relative in the odds: 4.8e-7 in the logarithm)


def _irregular_graph(n, s, seed):
    """Bits of degree 2 ... 5 on random checks (checks of degree ~3 ... 14), one check wider than any register bucket's
    straight-line code would like and one wide bit."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for j in range(n):
        for i in rng.choice(s, int(rng.integers(2, 6)), replace=False):
            rows.append(int(i)); cols.append(j)
    for j in rng.choice(n, 40, replace=False):
        rows.append(5); cols.append(int(j))
    for i in rng.choice(s, 19, replace=False):
        rows.append(int(i)); cols.append(9)
    for i in rng.choice(s, 6, replace=False):           # six checks of some 20 edges (the 16-wide bucket's two halves)
        for j in rng.choice(n, 13, replace=False):
            rows.append(int(i)); cols.append(int(j))
    H = sp.csc_matrix((np.ones(len(rows), dtype=np.uint8), (rows, cols)), shape=(s, n))
    H.sum_duplicates()
    H.data[:] = 1
    H.sort_indices()
    return H


@pytest.mark.parametrize("per", [0.10, 0.02])
def test_irregular_graph_keeps_whole_checks_in_lds(ldpc, gpu, per):
    """Rows on chip for an IRREGULAR graph of the C3 size (round 4; the reference takes any H, belief_propagation.jl:61-67):
    the host packs whole checks into the LDS of the workgroups that own them (team_irr_tables()), the team kernel's IRR
    instantiation updates them there.  16,384 syndromes through the default path against the tile kernel bit for bit
    (LLRs included) and against the oracle on a sample, above and below the graph's threshold."""
    n, s, batch = 16384, 8192, 16384
    H = _irregular_graph(n, s, seed=2024)
    cdeg = np.diff(H.tocsr().indptr)
    assert int(((cdeg > 16) & (cdeg <= 32)).sum()) >= 6 and int((cdeg > 32).sum()) == 1     # both wide-check paths of the IRR kernel
    syn_h = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, batch, per, seed=int(per * 1000)))
    syn = torch.from_numpy(syn_h).cuda()
    res = {}
    for variant in (0, 1):
        dec = ldpc.BeliefPropagationDecoder(H, per, ITERS, kernel_variant=variant, defer_threshold=0 if variant == 0 else -1)
        err = torch.empty((batch, n), dtype=torch.uint8, device="cuda"); conv = torch.empty(batch, dtype=torch.uint8, device="cuda")
        its = torch.empty(batch, dtype=torch.int32, device="cuda"); llr = torch.empty((batch, n), dtype=torch.float64, device="cuda")
        dec.decode_batch_device(syn, err, conv, llr, its)
        dec.last_status()
        info = dec.info()
        if variant == 0:
            assert info.last_kernel == 4 and info.last_team_size >= 16 and info.last_lds_rows >= 200, (info.last_kernel, info.last_team_size, info.last_lds_rows)
            assert info.last_rows_on_chip >= 0.08 * H.nnz, (info.last_rows_on_chip, H.nnz)
            # ... and once more without LLRs (the other instantiation)
            e2 = torch.empty_like(err); c2 = torch.empty_like(conv); i2 = torch.empty_like(its)
            dec.decode_batch_device(syn, e2, c2, None, i2)
            dec.last_status()
            assert torch.equal(e2, err) and torch.equal(c2, conv) and torch.equal(i2, its)
            del e2, c2, i2
        else:
            assert info.last_kernel == 1
        dec.close()
        res[variant] = (err, conv, its, llr.view(torch.int64))
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    err, conv, its, _ = res[0]
    conv_b = conv.bool()
    assert bool((its[~conv_b] == ITERS).all()) and bool((its >= 1).all()) and bool((its <= ITERS).all())
    idx = np.sort(np.random.default_rng(3).choice(batch, 384, replace=False))
    oerr, oconv, oits = _oracle_subset(H, per, syn_h[idx])
    assert np.array_equal(conv.cpu().numpy()[idx], oconv) and np.array_equal(its.cpu().numpy()[idx], oits)
    assert np.array_equal(err.cpu().numpy()[idx], oerr)
