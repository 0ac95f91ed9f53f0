#!/usr/bin/env python3
"""bench.py -- decoded syndromes/s of the MI355X BP decoder on BASELINE.json's metric.

One "step" = one batchdecode!-equivalent pass over one HBM-resident batch of synthetic
syndromes.  Default workload = BASELINE configs[2]: Gallager (4,8)-regular LDPC, n=16384,
m=8192, row weight 8, batch 65536 per GPU, max_iters 50, per = 0.10 ("full-50": above the BP
threshold, so every syndrome runs all 50 iterations).

N > 1 is BASELINE configs[3] as worded: ONE root-held (N x 65536) x s syndrome matrix
(belief_propagation.jl:220), contiguous shards scattered from GPU 0, decoded, hard decisions /
flags / iteration counts gathered on GPU 0 (weak scaling; no collective inside the decode --
syndromes are independent).  Two hosts for the same exchange:
  --mode scatter (default)  one process per GPU (torch.distributed.run), shards moved with
                            RCCL send/recv groups by ldpcdecoders.jl_amd/sharding.py;
  --mode cabi               ONE process drives all N GPUs through the C ABI
                            (ldpc_bp_create_multi / ldpc_bp_decode_batch_multi_device:
                            what a Julia host binds with ccall); other ranks, if the
                            launcher started any, only take part in the barriers.
N = 1 is the degenerate case of either (no exchange, decode straight into the result arrays).

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  roofline     : algorithmic bytes (32*nnz per syndrome*iteration) / sweep-kernel time
  cpu_baseline : the CPU oracle on edge lists over all host cores + its reference-faithful
                 dense mode on one, bounded samples, every sample checked against the GPU
  also         : (default workload, N = 1) the waterfall and the realistic error rate of the
                 same code, 3 steps each, under the same oracle gate.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n, wr, wc, batch per GPU, per, max_iters)
    "c3_full50": (16384, 8, 4, 65536, 0.10, 50),
    "c3_realistic": (16384, 8, 4, 65536, 0.02, 50),
    # the same code in the waterfall (threshold of the (4,8) ensemble ~0.076): broad iteration distribution,
    # thousands of stragglers for the hand-off
    "c3_waterfall": (16384, 8, 4, 65536, 0.06, 50),
    "c2_n1008": (1008, 6, 3, 4096, 0.01, 50),
    # the code of the reference's own tests and benchmark suite (test/test_bp_decoder.jl:7,
    # benchmark/benchmarks.jl:8-11): (9,10)-regular n=1000, per 0.01, 100 iterations
    "ref_1000_10_9": (1000, 10, 9, 65536, 0.01, 100),
    "ref_1000_10_9_hard": (1000, 10, 9, 65536, 0.06, 100),
    # a code whose messages fit a CU's LDS (128 KiB) but whose graph copy does not: node kernel, messages in LDS
    "mid_4096": (4096, 8, 4, 65536, 0.02, 50),
    # a large code with wider nodes (row weight 10, column weight 5): the 16-wide register bucket; rows on chip <10,5>
    "wide_16000_10_5": (16000, 10, 5, 32768, 0.10, 50),
    # a (3,6)-regular code of the C3 size (24 MiB a message slot: eight persistent teams; rows on chip <6,3>)
    "reg36_16380": (16380, 6, 3, 65536, 0.10, 50),
    # round 4: the other regular pairs of north_star's range keep their rows on chip too: (3,9) and (4,10) at the C3 size
    "reg39_16380": (16380, 9, 3, 65536, 0.10, 50),
    "reg410_16380": (16380, 10, 4, 32768, 0.10, 50),
    # BASELINE configs[4]: BB [[72,12,6]] H_X, BP on the GPU + OSD-0 on the host for what BP leaves
    "c5_bb72_bposd": (72, 6, 3, 1048576, 0.005, 50),
}
HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(workload, kernel="bp_tile_kernel"):
    """Bytes per launch of the sweep kernel between the L2s and the memory fabric, from the committed rocprofv3 PMC
    passes (profiles/*_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction applied).  bench.py
    cannot collect counters itself; the number is only reported for the workload AND kernel it was taken on."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("workload") == workload and t.get("kernel", "bp_tile_kernel") == kernel:
            best = t
    return (best["traffic_bytes_per_launch"], os.path.basename(best["source"].split(" ")[0])) if best else (None, None)


def make_syndromes(torch, H_csr, n, batch, per, seed, device):
    """Bernoulli(per) errors and their syndromes, generated on the device (synthetic data)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    indptr = torch.from_numpy(H_csr.indptr.astype(np.int64)).to(device)
    cols = torch.from_numpy(H_csr.indices.astype(np.int64)).to(device)
    s = H_csr.shape[0]
    deg = int(H_csr.indptr[1] - H_csr.indptr[0])
    regular = bool(np.all(np.diff(H_csr.indptr) == deg))
    syn = torch.empty((batch, s), dtype=torch.uint8, device=device)
    chunk = 4096
    for b0 in range(0, batch, chunk):
        b1 = min(batch, b0 + chunk)
        e = (torch.rand((b1 - b0, n), generator=g, device=device) < per).to(torch.uint8)
        if regular:
            syn[b0:b1] = e[:, cols].view(b1 - b0, s, deg).sum(dim=2, dtype=torch.int32).remainder(2).to(torch.uint8)
        else:
            ge = e[:, cols].to(torch.int32)
            cs = torch.cumsum(ge, dim=1)
            cs = torch.cat([torch.zeros((b1 - b0, 1), dtype=cs.dtype, device=device), cs], dim=1)
            syn[b0:b1] = (cs[:, indptr[1:]] - cs[:, indptr[:-1]]).remainder(2).to(torch.uint8)
    return syn


def cpu_baseline_dense(H, per, max_iters, syn_sample, gpu_err, gpu_conv, budget_s=12.0):
    """Oracle in reference-faithful DENSE mode (two dense s x n Float64 matrices, full reset
    per decode, strided access: the cost structure of belief_propagation.jl:83-91,121-188),
    single thread like the reference.  Also re-checks the GPU result on the sample."""
    from oracle import BPOracle

    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=max_iters, dense=True)
    done, t0, ok = 0, time.perf_counter(), True
    for b in range(syn_sample.shape[0]):
        err, conv = oc.decode(syn_sample[b])
        ok = ok and bool(conv) == bool(gpu_conv[b]) and np.array_equal(err.astype(np.uint8), gpu_err[b])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, ok


def cpu_baseline_edge_list(H, per, max_iters, syn_sample, gpu_err, gpu_conv, gpu_iters, threads, budget_s=20.0):
    """The same arithmetic on the structural non-zeros only (the strongest honest CPU figure: no dense
    reset!, no strided access), one oracle decoder per host thread (the reference decoder is not
    re-entrant; the C code runs without the GIL).  Every decoded syndrome is compared with the GPU's
    hard decisions, flag and iteration count."""
    import concurrent.futures as cf

    from oracle import BPOracle

    def work(idx):
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=max_iters)
        t0, done, ok = time.perf_counter(), 0, True
        for b0 in range(0, len(idx), 16):
            sl = idx[b0:b0 + 16]
            err, conv, _, its = oc.batchdecode(syn_sample[sl], want_llr=False)
            ok = ok and np.array_equal(err, gpu_err[sl]) and np.array_equal(conv, gpu_conv[sl]) and np.array_equal(its, gpu_iters[sl])
            done += len(sl)
            if time.perf_counter() - t0 > budget_s:
                break
        return done, ok

    chunks = np.array_split(np.arange(syn_sample.shape[0]), threads)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(work, chunks))
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    return done / wall, done, all(r[1] for r in res)


# What tools/mall_probe.hip streams in place (read-modify-write of 512-byte rows) through the XCDs' fabric ports when
# the regions fit the Infinity Cache, by the number of XCDs that host a region (profiles/r02_infinity_cache_probe.txt;
# measured ceilings, not guide figures)
CACHE_SIDE_CEILING_GBS = {4: 5920.0, 5: 7250.0, 6: 7390.0, 7: 8760.0, 8: 11010.0}
INFINITY_CACHE_BYTES = 256 << 20


def roofline_object(dec, args, kname, last_kernel, nnz, s_checks, n, batch, achieved, alg_bytes, sweep_ms, total_ms):
    """`achieved` / `peak` / `frac` are the metric's own yardstick: algorithmic message bytes over the sweep kernel's
    time against the HBM peak (BASELINE.json: "achieved HBM GB/s fraction").  `bound` says what actually limits the
    kernel that ran: the persistent teams keep their message slots inside the Infinity Cache, so their bound is the
    cache side of the fabric, and `bound_ceiling` sets the (committed, PMC-measured) bytes between the L2s and the
    fabric against the measured in-place streaming rate of as many XCDs' ports; `hbm_side` bounds what the HBM itself
    has to move for such a launch from below (syndromes in, results out)."""
    info = dec.info()
    default_size = not args.batch and not args.per
    traffic, traffic_src = pmc_traffic(args.workload, kname) if default_size else (None, None)
    slots = int(info.resident_tiles // max(info.last_team_size, 1)) if last_kernel == 4 else 0
    rows_on_chip = int(getattr(info, "last_rows_on_chip", 0)) if last_kernel == 4 else 0   # rows of a tile in LDS / registers: not in the slot
    slot_bytes = slots * (nnz - rows_on_chip) * 512
    in_cache = last_kernel == 4 and 0 < slot_bytes <= INFINITY_CACHE_BYTES
    on_chip = bool(last_kernel == 2 or (last_kernel == 3 and nnz * 8 <= 150 * 1024))
    ticks = dec.phase_ticks(0)
    obj = {
        "bound": "infinity_cache" if in_cache else "hbm",
        "kernel": kname,
        # the LDS-resident kernels keep the messages on chip: their "achieved" is the algorithmic message
        # traffic they would have cost in HBM, not bytes the HBM moved (it can exceed the peak)
        "messages_on_chip": on_chip,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "peak_is": "HBM3E peak of the MI355X (the yardstick BASELINE.json's metric names), not this kernel's bound" if in_cache else "HBM3E peak of the MI355X",
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        # NOT measured in this run: bench.py cannot collect PMC counters; this is the committed rocprofv3
        # figure (profiles/*_traffic.json) of the same workload AND kernel, named in traffic_source
        "traffic": traffic,
        "traffic_source": traffic_src,
        "message_slots_in_flight_bytes": slot_bytes if last_kernel == 4 else None,   # what the teams keep in the Infinity Cache
        "message_rows_on_chip_frac": (rows_on_chip / nnz) if last_kernel == 4 else None,   # ... and what never leaves the CUs
        "alg_bytes_per_launch": alg_bytes,
        "kernel_ms": sweep_ms,
        "pack_sweep_unpack_ms": total_ms,
        "phase_share_check_var_conv": [round(x / max(sum(ticks), 1), 4) for x in ticks],
    }
    if in_cache:
        ceiling = CACHE_SIDE_CEILING_GBS.get(slots if slots <= 8 else 8, CACHE_SIDE_CEILING_GBS[8])
        fabric = traffic / (sweep_ms * 1e-3) / 1e9 if (traffic and sweep_ms > 0) else None
        # the honest fraction: real bytes through the fabric against what the fabric's cache side streams -- lead with this one;
        # `frac` above is BASELINE.json's yardstick (algorithmic bytes against an HBM peak these bytes never reach)
        obj["frac_of_bound"] = fabric / ceiling if fabric else None
        obj["bound_ceiling"] = {
            "what": f"in-place streaming of {min(slots, 8)} cache-resident regions, one per XCD (tools/mall_probe.hip, profiles/r02_infinity_cache_probe.txt)",
            "peak": ceiling, "unit": "GB/s",
            "achieved": fabric,        # committed PMC bytes between the L2s and the fabric / this run's kernel time
            "frac": fabric / ceiling if fabric else None,
        }
        io_bytes = float(batch) * (s_checks + n + 1 + 4)     # what must cross the HBM at least: syndromes in, results out
        obj["hbm_side"] = {"bytes_per_launch_lower_bound": io_bytes,
                           "GBs_lower_bound": io_bytes / (total_ms * 1e-3) / 1e9 if total_ms > 0 else None,
                           "note": "no counter in profiles/ separates HBM from Infinity-Cache hits (FETCH_SIZE / WRITE_SIZE count L2<->fabric requests)"}
    return obj


def llr_gate(H, per, max_iters, h_syn, h_llr, k=48):
    """max |LLR difference| against the oracle on the first k syndromes (+-Inf must match exactly: -> inf if not)."""
    from oracle import BPOracle

    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=max_iters)
    _, _, ollr, _ = oc.batchdecode(h_syn[:k], want_llr=True)
    g = h_llr[:k]
    fin = np.isfinite(ollr)
    if not np.array_equal(g[~fin], ollr[~fin]):
        return float("inf")
    return float(np.max(np.abs(g[fin] - ollr[fin]))) if fin.any() else 0.0


def also_workloads(ldpc, torch, H, Hcsr, n, nnz, device, local_rank, names=("c3_waterfall", "c3_realistic", "c3_full50_llr", "c3_realistic_llr"), steps=3, sample=512):
    """The same code at other error rates, after the timed region (they are not the headline; they make the
    early-exit path -- straggler hand-off, packed levels -- observable in the driver's own run): `steps` calls each,
    kernel time and iteration sum from the library's HIP events, and the same oracle gate on a sample.  The `_llr`
    entries ask for the LLRs of every syndrome (the reference's decode! always fills scratch.log_probabs,
    belief_propagation.jl:163; BP+OSD reads them): the instantiation of the sweep kernel that writes them, held to
    <= 1e-5 against the oracle (+-Inf exact) on a sample."""
    res = {}
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    for name in names:
        want_llr = name.endswith("_llr")
        _, _, _, batch, per, max_iters = WORKLOADS[name[:-4] if want_llr else name]
        dec = ldpc.BeliefPropagationDecoder(H, per, max_iters, device=local_rank)
        syn = make_syndromes(torch, Hcsr, n, batch, per, seed=1234, device=device)
        err = torch.empty((batch, n), dtype=torch.uint8, device=device)
        conv = torch.empty(batch, dtype=torch.uint8, device=device)
        iters = torch.empty(batch, dtype=torch.int32, device=device)
        llr = torch.empty((batch, n), dtype=torch.float64, device=device) if want_llr else None
        dec.decode_batch_device(syn, err, conv, llr, iters)      # warm-up (workspace, levels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            dec.decode_batch_device(syn, err, conv, llr, iters)
        dec.last_status()
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) / steps * 1e3
        per_call = [dec.last_timing(i) for i in range(steps)]
        kernel_ms = sum(x[0] for x in per_call) / steps
        total_ms = sum(x[1] for x in per_call) / steps
        sum_iters = per_call[0][2]
        gbs = sum_iters * 32.0 * nnz / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        k = min(batch, sample)
        h = [x[:k].cpu().numpy() for x in (syn, err, conv, iters)]
        _, done, ok = cpu_baseline_edge_list(H, per, max_iters, h[0], h[1], h[2], h[3], cores, budget_s=10.0)
        res[name] = {"per": per, "batch": batch, "steps": steps, "ms_per_step": wall_ms, "kernel_ms": kernel_ms, "pack_sweep_unpack_ms": total_ms,
                     "mean_iters": sum_iters / batch, "converged_frac": float(conv.float().mean().item()),
                     "value": batch / (wall_ms * 1e-3), "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                     "frac_is": "sum over syndromes of iterations executed x 32 nnz bytes / sweep-kernel time / HBM peak",
                     "kernel": {1: "bp_tile_kernel", 2: "bp_lds_kernel", 3: "bp_node_kernel", 4: "bp_team_kernel"}.get(int(dec.info().last_kernel), "?"),
                     "gpu_matches_oracle_on_sample": bool(ok), "oracle_sample": int(done)}
        if want_llr:
            kl = min(batch, 48)
            res[name]["llr_max_abs_diff_vs_oracle"] = llr_gate(H, per, max_iters, h[0], llr[:kl].cpu().numpy(), kl)
            res[name]["llr_sample"] = kl
            res[name]["gpu_matches_oracle_on_sample"] = bool(ok and res[name]["llr_max_abs_diff_vs_oracle"] <= 1e-5)
        dec.close()
        del syn, err, conv, iters, llr
    return res


def also_small_configs(ldpc, torch, device, local_rank):
    """BASELINE configs 1, 2 and 5 in the driver's own run, a few steps each, every result held against the oracle:
      c2_n1008_batch4096   (3,6)-regular n = 1008, per 0.01, batchdecode! of 4096 HBM-resident syndromes (LDS-resident kernel);
      c1_single_decode     the same code, batch = 1 through the HOST entry ldpc_bp_decode_batch with LLRs -- what a Julia
                           `decode!` costs (BASELINE config 1 names the reference's CPU path for this case; the CPU oracle is
                           timed beside it);
      c5_bb72_bposd        BB [[72,12,6]] H_X, per 0.005, 2^20 syndromes: BP on the GPU, OSD-0 on the host for what BP
                           leaves unconverged (ldpc_osd_postprocess_batch), end to end per step."""
    import scipy.sparse as sp

    from oracle import BPOracle, osd_oracle_postprocess

    res = {}
    # ---- config 2 and config 1: (3,6)-regular n = 1008
    n, wr, wc, batch, per, max_iters = WORKLOADS["c2_n1008"]
    H = ldpc.codes.parity_check_csc(n, wr, wc)
    Hcsr = H.tocsr()
    dec = ldpc.BeliefPropagationDecoder(H, per, max_iters, device=local_rank)
    syn = make_syndromes(torch, Hcsr, n, batch, per, seed=4321, device=device)
    err = torch.empty((batch, n), dtype=torch.uint8, device=device)
    conv = torch.empty(batch, dtype=torch.uint8, device=device)
    iters = torch.empty(batch, dtype=torch.int32, device=device)
    for _ in range(3):
        dec.decode_batch_device(syn, err, conv, None, iters)
    torch.cuda.synchronize()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        dec.decode_batch_device(syn, err, conv, None, iters)
    dec.last_status()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / steps * 1e3
    kernel_ms = sum(dec.last_timing(i)[0] for i in range(8)) / 8
    h_syn, h_err, h_conv, h_it = (t.cpu().numpy() for t in (syn, err, conv, iters))
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=max_iters)
    t0 = time.perf_counter()
    oerr, oconv, ollr, oits = oc.batchdecode(h_syn, want_llr=True)
    cpu_s = time.perf_counter() - t0
    ok = np.array_equal(oerr, h_err) and np.array_equal(oconv, h_conv) and np.array_equal(oits, h_it)
    res["c2_n1008_batch4096"] = {"per": per, "batch": batch, "steps": steps, "ms_per_step": wall_ms, "kernel_ms": kernel_ms,
                                 "value": batch / (wall_ms * 1e-3), "unit": "syndromes/s", "mean_iters": float(h_it.mean()),
                                 "converged_frac": float(h_conv.mean()), "kernel": {1: "bp_tile_kernel", 2: "bp_lds_kernel", 3: "bp_node_kernel", 4: "bp_team_kernel"}.get(int(dec.info().last_kernel), "?"),
                                 "gpu_matches_oracle_on_sample": bool(ok), "oracle_sample": int(batch),
                                 "cpu_oracle_1_thread_syndromes_per_s": batch / cpu_s}
    # config 1: one syndrome at a time through the host-pointer entry, LLRs included (decode! fills scratch.log_probabs)
    k = 64
    lat, ok1, worst = [], True, 0.0
    for b in range(k):
        e1, c1, l1, i1 = dec.decode_batch_host(h_syn[b:b + 1], want_llr=True, want_iters=True)   # (first calls: warm-up, kept out below)
        t0 = time.perf_counter()
        e1, c1, l1, i1 = dec.decode_batch_host(h_syn[b:b + 1], want_llr=True, want_iters=True)
        lat.append(time.perf_counter() - t0)
        ok1 = ok1 and np.array_equal(e1[0], oerr[b]) and c1[0] == oconv[b] and i1[0] == oits[b]
        fin = np.isfinite(ollr[b])
        ok1 = ok1 and np.array_equal(l1[0][~fin], ollr[b][~fin])
        worst = max(worst, float(np.max(np.abs(l1[0][fin] - ollr[b][fin]))) if fin.any() else 0.0)
    t0 = time.perf_counter()
    for b in range(k):
        oc.decode(h_syn[b])
    cpu_us = (time.perf_counter() - t0) / k * 1e6
    res["c1_single_decode"] = {"what": "decode! = ldpc_bp_decode_batch(batch = 1, host buffers, LLRs) on the (3,6) n = 1008 code, per 0.01, 50 iterations at most",
                               "calls": k, "us_per_decode_median": float(np.median(lat) * 1e6), "us_per_decode_mean": float(np.mean(lat) * 1e6),
                               "value": 1.0 / float(np.median(lat)), "unit": "syndromes/s",
                               "gpu_matches_oracle_on_sample": bool(ok1 and worst <= 1e-5), "llr_max_abs_diff_vs_oracle": worst,
                               "cpu_oracle_edge_list_us_per_decode": cpu_us}
    dec.close()
    del syn, err, conv, iters
    # ---- config 5: BB [[72,12,6]], BP + OSD-0 on the host
    n5, _, _, batch5, per5, it5 = WORKLOADS["c5_bb72_bposd"]
    Hd = ldpc.codes.bivariate_bicycle_72_12_6()[0]
    H5 = sp.csc_matrix(Hd)
    H5.sort_indices()
    bposd = ldpc.BeliefPropagationOSDDecoder(H5, per5, it5, osd_order=0, device=local_rank)
    syn5 = make_syndromes(torch, H5.tocsr(), n5, batch5, per5, seed=555, device=device)
    e5, c5, k5 = bposd.batchdecode_device(syn5)
    torch.cuda.synchronize()
    steps5 = 5
    t0 = time.perf_counter()
    for _ in range(steps5):
        e5, c5, k5 = bposd.batchdecode_device(syn5)
    torch.cuda.synchronize()
    ms5 = (time.perf_counter() - t0) / steps5 * 1e3
    # gate: every syndrome BP left unconverged (what OSD touched) and the first 2048 others, against the oracle chain
    # BP oracle -> OSD oracle (dense restatement of belief_propagation_osd.jl:52-60, :63-125)
    hc = c5.cpu().numpy()
    pick = np.unique(np.concatenate([np.flatnonzero(hc == 0)[:512], np.arange(2048)]))
    hs, he = syn5[torch.from_numpy(pick).to(device)].cpu().numpy(), e5[torch.from_numpy(pick).to(device)].cpu().numpy()
    oc5 = BPOracle(csc=(H5.indptr, H5.indices), shape=H5.shape, per=per5, max_iters=it5)
    o_err, o_conv, o_llr, _ = oc5.batchdecode(hs, want_llr=True)
    ok5 = np.array_equal(o_conv, hc[pick])
    satisfied = True
    for q in range(len(pick)):
        want = o_err[q] if o_conv[q] else osd_oracle_postprocess(Hd, hs[q], o_err[q], o_llr[q], 0)
        ok5 = ok5 and np.array_equal(want, he[q])
        satisfied = satisfied and np.array_equal((Hd.astype(np.int64) @ he[q].astype(np.int64)) % 2, hs[q])   # test_bposd_decoder.jl:37-47
    res["c5_bb72_bposd"] = {"per": per5, "batch": batch5, "steps": steps5, "ms_per_step": ms5, "value": batch5 / (ms5 * 1e-3), "unit": "syndromes/s",
                            "converged_frac": float(hc.mean()), "osd_postprocessed_per_step": int(k5),
                            "what": "BP on the GPU for all, the unconverged ones once more with LLRs, OSD-0 for those on the host (ldpc_osd_postprocess_batch), results back in HBM",
                            "gpu_matches_oracle_on_sample": bool(ok5), "oracle_sample": int(len(pick)), "output_satisfies_syndrome_on_sample": bool(satisfied)}
    bposd.bp_decoder.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3_full50", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="syndromes per GPU (default: the workload's)")
    ap.add_argument("--per", type=float, default=0.0, help="physical error rate (default: the workload's)")
    ap.add_argument("--mode", default="auto", choices=["auto", "scatter", "cabi", "replicas"],
                    help="scatter (default): BASELINE config 4 as worded -- rank 0 holds the whole N x batch matrix in HBM, "
                         "ldpcdecoders.jl_amd.sharding scatters the shards over RCCL, every rank decodes, the root gathers; "
                         "N = 1 is the degenerate case of the same code.  cabi: the same exchange from ONE process through the "
                         "C ABI (ldpc_bp_create_multi: one handle and stream per GPU, RCCL send/recv from GPU 0).  replicas: every "
                         "rank decodes a shard of its own, no exchange (round 1's mode)")
    ap.add_argument("--no-also", action="store_true", help="skip the extra workloads of the `also` object")
    ap.add_argument("--verify", action="store_true",
                    help="scatter mode: after the timed steps the root decodes every shard once more by itself and compares")
    ap.add_argument("--waves-per-tile", type=int, default=0)
    ap.add_argument("--resident-tiles", type=int, default=0)
    ap.add_argument("--kernel-variant", type=int, default=0, help="0 auto, 1 HBM-streaming, 2 LDS-resident, 3 node-parallel, 4 team")
    ap.add_argument("--defer-threshold", type=int, default=0, help="0 auto (16), -1 off (streaming kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--llr", action="store_true",
                    help="the timed step asks for the LLRs of every syndrome as well (scratch.log_probabs, belief_propagation.jl:163; "
                         "one GPU, decode straight into the result arrays)")
    ap.add_argument("--llr-exact", action="store_true", help="with --llr: ldpc_bp_options.llr_exact (LLRs from the full posterior odds)")
    ap.add_argument("--prealloc-gib", type=float, default=0.0,
                    help="experiment: hold this much HBM before anything else is allocated (shifts physical placement)")
    # rehearsal of the N>1 control flow on a box with fewer GPUs than ranks (never used by the driver):
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share cuda:0; the exchange runs over gloo with the shards staged through the host instead of RCCL")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cabi = args.mode == "cabi"
    if world != args.gpus and not (cabi and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.rehearse_on_one_gpu or cabi:
        local_rank = 0                      # (cabi: rank 0's process drives every GPU; the batch lives on GPU 0)
    if cabi and rank != 0:
        # a rank the launcher started but the single-process host does not need: it joins the barriers and leaves
        dist.init_process_group("gloo")
        for _ in range(3):
            dist.barrier()
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu or cabi:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)   # RCCL over xGMI
    shards = args.gpus                      # shards of the job = GPUs (logical ones in a rehearsal)

    import ldpcdecoders_jl_amd as ldpc
    from ldpcdecoders_jl_amd import sharding

    hold = torch.empty(int(args.prealloc_gib * (1 << 30)), dtype=torch.uint8, device=device) if args.prealloc_gib > 0 else None
    n, wr, wc, batch, per, max_iters = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    if args.per > 0:
        per = args.per
    bposd_workload = args.workload == "c5_bb72_bposd"
    mode = args.mode
    if mode == "auto":
        mode = "replicas" if (bposd_workload or args.llr) else "scatter"
    if args.llr and (mode != "replicas" or bposd_workload):
        raise SystemExit("--llr: --mode replicas (or auto) on a plain BP workload")
    if bposd_workload and mode == "scatter":
        raise SystemExit("the BP+OSD workload has a host step per rank: use --mode replicas")
    if bposd_workload:
        import scipy.sparse as sp

        H = sp.csc_matrix(ldpc.codes.bivariate_bicycle_72_12_6()[0])
        H.sort_indices()
    else:
        H = ldpc.codes.parity_check_csc(n, wr, wc)
    nnz = int(H.nnz)
    s_checks = int(H.shape[0])
    if cabi:
        cabi_devices = [0] * shards if args.rehearse_on_one_gpu else list(range(shards))
        dec = ldpc.BeliefPropagationDecoder(H, per, max_iters, devices=cabi_devices,
                                            waves_per_tile=args.waves_per_tile, resident_tiles=args.resident_tiles,
                                            kernel_variant=args.kernel_variant, defer_threshold=args.defer_threshold)
    else:
        dec = ldpc.BeliefPropagationDecoder(H, per, max_iters, device=local_rank,
                                            waves_per_tile=args.waves_per_tile, resident_tiles=args.resident_tiles,
                                            kernel_variant=args.kernel_variant, defer_threshold=args.defer_threshold,
                                            llr_exact=args.llr_exact)
    # Synthetic data: shard g of the job is Bernoulli(per) errors from seed 1234 + g.  scatter mode: the root holds
    # all N shards as ONE caller-owned matrix (belief_propagation.jl:220: one `syndromes`, columns independent).
    Hcsr = H.tocsr()
    if mode in ("scatter", "cabi"):
        syn = None
        if rank == 0:
            syn = torch.empty((batch * shards, s_checks), dtype=torch.uint8, device=device)
            for g in range(shards):
                syn[g * batch:(g + 1) * batch] = make_syndromes(torch, Hcsr, n, batch, per, seed=1234 + g, device=device)
        err = conv = iters = None
        if cabi:
            err = torch.empty((batch * shards, n), dtype=torch.uint8, device=device)
            conv = torch.empty(batch * shards, dtype=torch.uint8, device=device)
            iters = torch.empty(batch * shards, dtype=torch.int32, device=device)
    else:
        syn = make_syndromes(torch, Hcsr, n, batch, per, seed=1234 + rank, device=device)
        err = torch.empty((batch, n), dtype=torch.uint8, device=device)
        conv = torch.empty(batch, dtype=torch.uint8, device=device)
        iters = torch.empty(batch, dtype=torch.int32, device=device)
    llr_out = torch.empty((batch, n), dtype=torch.float64, device=device) if args.llr else None

    osd_sent = [0]
    device_calls = [0]   # library calls made by the timed steps (1 per step unless noted)
    phases = []          # scatter mode: per timed step {scatter_ms, decode_ms, gather_ms} of this rank
    result = [None]
    bposd = None
    if bposd_workload:
        bposd = ldpc.BeliefPropagationOSDDecoder(H, per, max_iters, osd_order=0, device=local_rank,
                                                 waves_per_tile=args.waves_per_tile, kernel_variant=args.kernel_variant)
        dec.close()
        dec = bposd.bp_decoder

        def step():
            e, c, k = bposd.batchdecode_device(syn)
            err.copy_(e)
            conv.copy_(c)
            osd_sent[0] = k
            device_calls[0] += 2 if k else 1   # BP for all, then the unconverged ones once more with LLRs
    elif cabi:
        def step():
            dec.decode_batch_device(syn, err, conv, None, iters)   # ldpc_bp_decode_batch_multi_device: scatter, decode, gather
            result[0] = (err, conv, iters)
    elif mode == "scatter":
        decode_fn = sharding.gpu_decode_fn(dec)
        comm = torch.device("cpu") if (args.rehearse_on_one_gpu and world > 1) else None

        def step():
            tm = {}
            result[0] = sharding.batchdecode_sharded(decode_fn, syn, s_checks, n, root=0, device=device,
                                                     comm_device=comm, timing=tm)
            phases.append(tm)
    else:
        def step():
            dec.decode_batch_device(syn, err, conv, llr_out, iters)

    def fence():
        if cabi:
            dec.last_status()               # every device's share of the calls enqueued so far
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if mode == "scatter":
        # batchdecode_sharded returns NEW result arrays every call (the reference's batchdecode! takes the caller's; the
        # sharded entry has none to take): two generations are alive across the assignment above, so torch's caching allocator
        # needs two blocks of that size -- a hipMalloc of 1 GiB inside one of a handful of timed steps showed as +8 ms per
        # step at per 0.02 (59.5 against 51.7 ms, kernel time unchanged).  Primed here, outside the timed region.
        prime = [torch.empty((batch * shards if rank == 0 else batch, n), dtype=torch.uint8, device=device) for _ in range(2)]
        del prime
    for _ in range(args.warmup):
        step()
    fence()
    device_calls[0] = 0
    del phases[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # HIP-event times of the sweep kernel of the timed steps (the library keeps the events of
    # its last 16 calls, recorded on the launch stream), read after the timed region
    last_kernel = int(dec.info().last_kernel)
    kname = {1: "bp_tile_kernel", 2: "bp_lds_kernel", 3: "bp_node_kernel", 4: "bp_team_kernel"}.get(last_kernel, "?")
    cps = max(1, device_calls[0] // max(args.steps, 1)) if device_calls[0] else 1   # calls per step
    k = max(1, min(args.steps, 16 // cps))
    per_call = [dec.last_timing(i) for i in range(k * cps)]
    sweep_ms = sum(t[0] for t in per_call) / k
    total_ms = sum(t[1] for t in per_call) / k
    sum_iters = sum(t[2] for t in per_call[:cps])
    ph = {key: (sum(p[key] for p in phases) / len(phases) if phases else 0.0) for key in ("scatter_ms", "decode_ms", "gather_ms")}
    decode_ms_max = ph["decode_ms"]
    if cabi:
        mi = dec.multi_info()
        ph = {"scatter_ms": mi.scatter_ms, "decode_ms": mi.decode_ms_max, "gather_ms": mi.gather_ms}
        decode_ms_max = mi.decode_ms_max
    if world > 1 and cabi:
        dist.barrier()
    elif world > 1:
        cdev = "cpu" if args.rehearse_on_one_gpu else device
        t = torch.tensor([elapsed, ph["decode_ms"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, decode_ms_max = float(t[0].item()), float(t[1].item())

    verified = None
    if mode in ("scatter", "cabi") and args.verify and rank == 0:
        # the root decodes every shard by itself and compares with what came back from the ranks
        g_err, g_conv, g_it = result[0]
        e1 = torch.empty((batch, n), dtype=torch.uint8, device=device)
        c1 = torch.empty(batch, dtype=torch.uint8, device=device)
        i1 = torch.empty(batch, dtype=torch.int32, device=device)
        verified = True
        vdec = ldpc.BeliefPropagationDecoder(H, per, max_iters, device=local_rank) if cabi else dec
        for g in range(shards):
            vdec.decode_batch_device(syn[g * batch:(g + 1) * batch], e1, c1, None, i1)
            torch.cuda.synchronize()
            sl = slice(g * batch, (g + 1) * batch)
            verified = verified and bool(torch.equal(e1, g_err[sl])) and bool(torch.equal(c1, g_conv[sl])) and bool(torch.equal(i1, g_it[sl]))
        if cabi:
            vdec.close()

    if rank == 0:
        if mode in ("scatter", "cabi"):
            err, conv, iters = (t[:batch] for t in result[0])   # the root's own shard (seed 1234)
            syn0 = syn[:batch]
        else:
            syn0 = syn
        total_syndromes = batch * shards * args.steps
        value = total_syndromes / elapsed
        alg_bytes = float(sum_iters) * 32.0 * nnz        # 32*nnz B per syndrome*iteration
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        if cabi:
            xk = {0: "auto", 1: "hipMemcpyPeerAsync (logical devices share a GPU: rehearsal)", 2: "RCCL ncclSend/ncclRecv groups over xGMI", 3: "none"}[int(mi.exchange)]
            par = (f"ONE process, C ABI (ldpc_bp_create_multi): one root-held {batch * shards} x {s_checks} syndrome matrix on GPU 0 -> "
                   f"contiguous shards to {shards} devices {list(mi.devices[:mi.ndev])}, exchange: {xk} -> one handle + stream per device "
                   f"-> results gathered on GPU 0" if shards > 1 else
                   "ldpc_bp_decode_batch_multi_device with one device (degenerate case: no exchange, decode straight into the result arrays)")
        elif mode == "scatter":
            par = (f"one root-held {batch * world} x {s_checks} syndrome matrix -> contiguous shards scattered from rank 0 "
                   f"({'gloo via host staging, all ranks on one GPU (rehearsal)' if args.rehearse_on_one_gpu and world > 1 else 'RCCL send/recv groups over xGMI'}) "
                   f"-> {world} x HIP decode -> hard decisions / flags / iteration counts gathered on rank 0"
                   if world > 1 else
                   "sharding.batchdecode_sharded at world size 1 (degenerate case: no exchange, decode straight into the result arrays)")
        else:
            par = f"replicas x{world}: every rank decodes a shard of its own (independent syndromes, no exchange)"
        out = {
            "metric": "decoded syndromes/sec (batchdecode!, 50 iters) + achieved HBM GB/s fraction",
            "value": value,
            "unit": "syndromes/s",
            "n_gpus": shards,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: Gallager ({wc},{wr})-regular LDPC n={n} m={s_checks} nnz={nnz}, "
                            f"batch={batch}/GPU, per={per}, max_iters={max_iters}, HBM-resident uint8 syndromes in, "
                            f"uint8 hard decisions + converged + iteration counts out" + (" + f64 LLRs of every syndrome" if args.llr else ""),
                "global_batch": batch * shards,
                "mode": mode,
                "parallelism": par,
                "mean_iters": sum_iters / batch,
                "converged_frac": float(conv.float().mean().item()),
                "osd_postprocessed_per_step": osd_sent[0],
            },
            "roofline": roofline_object(dec, args, kname, last_kernel, nnz, s_checks, n, batch, achieved, alg_bytes, sweep_ms, total_ms),
        }
        if mode in ("scatter", "cabi"):
            out["exchange"] = {"scatter_ms": ph["scatter_ms"], "decode_ms_max_rank": decode_ms_max, "gather_ms": ph["gather_ms"],
                               "scatter_bytes_per_peer": batch * s_checks if shards > 1 else 0,
                               "gather_bytes_per_peer": batch * (n + 1 + 4) if shards > 1 else 0,
                               "sharded_matches_local": verified}
        if not args.no_cpu_baseline and shards == 1 and not bposd_workload:
            cores = max(1, min(len(os.sched_getaffinity(0)), 16))   # a one-GPU box grants 16 cores
            k_edge = min(batch, 4096)       # (the gate of a reported C3 number is >= 4,096 syndromes)
            h_syn, h_err, h_conv, h_it = (t[:k_edge].cpu().numpy() for t in (syn0, err, conv, iters))
            e_rate, e_done, e_ok = cpu_baseline_edge_list(H, per, max_iters, h_syn, h_err, h_conv, h_it, cores)
            k_dense = min(batch, 64)
            d_rate, d_done, d_ok = cpu_baseline_dense(H, per, max_iters, h_syn[:k_dense], h_err[:k_dense], h_conv[:k_dense])
            out["cpu_baseline"] = {
                "value": e_rate, "unit": "syndromes/s", "cores": cores, "kind": "port",
                "sample": f"first {e_done} syndromes of the same batch, C oracle on edge lists (the reference's arithmetic on the "
                          f"structural non-zeros only), one decoder per thread on {cores} host threads; "
                          f"the Julia reference itself cannot run here (no Julia runtime)",
                "gpu_matches_oracle_on_sample": bool(e_ok and d_ok),
                **({"llr_max_abs_diff_vs_oracle": llr_gate(H, per, max_iters, h_syn, llr_out[:48].cpu().numpy(), min(batch, 48))} if args.llr else {}),
                "reference_faithful_1_thread": {
                    "value": d_rate, "unit": "syndromes/s", "cores": 1,
                    "sample": f"first {d_done} syndromes, C oracle in reference-faithful dense mode (2 dense {s_checks}x{n} Float64 "
                              f"matrices, full reset! per decode: the cost structure of belief_propagation.jl:83-91,121-188), "
                              f"1 thread like the reference",
                },
            }
        if args.workload == "c3_full50" and shards == 1 and mode != "replicas" and not args.no_also and not args.batch and not args.per \
                and not args.kernel_variant and not args.waves_per_tile and not args.resident_tiles:
            out["also"] = also_workloads(ldpc, torch, H, Hcsr, n, nnz, device, local_rank)
            out["also"].update(also_small_configs(ldpc, torch, device, local_rank))
        print(json.dumps(out), flush=True)
    # explicit teardown, in this order, before the interpreter starts dismantling modules: the decoder handles
    # (hipDeviceSynchronize + frees), then the process group
    result[0] = None
    if bposd is not None and hasattr(bposd, "close"):
        bposd.close()
    dec.close()
    del hold
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
