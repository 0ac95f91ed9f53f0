#!/usr/bin/env python3
"""CPU baselines B1..B6 of BASELINE.md section 3, measured on the host cores of the box it runs on.

The literal reference (Julia) cannot run here; these are the C oracle's two storage modes:
  reference-faithful = dense s x n Float64 matrices, full reset! per decode, strided access
                       (the cost structure of src/decoders/belief_propagation.jl:83-91,121-188)
  edge-list          = same arithmetic on the structural non-zeros only
one decoder per thread (the reference decoder is not re-entrant), threads through ctypes (the
C code runs without the GIL).  Prints a markdown table."""
import concurrent.futures as cf
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpcdecoders_jl_amd as ldpc  # noqa: E402  (host-side generators only; no GPU call is made)
from oracle import BPOracle, osd_oracle_postprocess  # noqa: E402

NCPU = min(os.cpu_count() or 1, 16)   # a one-GPU box grants 16 cores


def rate(H, per, iters, syn, dense, threads, budget_s=25.0):
    """syndromes/s; every thread decodes its share until the time budget is used up."""
    def work(chunk):
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=iters, dense=dense)
        t0, done = time.perf_counter(), 0
        step = 1 if dense else 64          # the C batch loop (bp_oracle_decode_batch), no Python per syndrome
        for b in range(0, chunk.shape[0], step):
            oc.batchdecode(chunk[b:b + step], want_llr=False)
            done += min(step, chunk.shape[0] - b)
            if time.perf_counter() - t0 > budget_s:
                break
        return done, time.perf_counter() - t0

    chunks = np.array_split(syn, threads)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(work, chunks))
    wall = time.perf_counter() - t0
    return sum(r[0] for r in res) / wall, sum(r[0] for r in res)


def main():
    rows = []
    # ---- C1/C2: (3,6)-regular n=1008, per 0.01, 50 iterations
    H = ldpc.codes.parity_check_csc(1008, 6, 3)
    S = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(1008, 4096, 0.01, seed=1))
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.01, max_iters=50, dense=True)
    ts = []
    for b in range(1000):
        t0 = time.perf_counter(); oc.decode(S[b]); ts.append(time.perf_counter() - t0)
    rows.append(("B1", "C1 n=1008, batch 1 (`decode!`)", "reference-faithful", 1, f"median {np.median(ts)*1e6:.0f} µs/decode over 1000 syndromes"))
    r, k = rate(H, 0.01, 50, S, True, 1)
    rows.append(("B2", "C2 n=1008, batch 4096", "reference-faithful", 1, f"{r:,.0f} syndromes/s ({k} decoded)"))
    for th in (1, NCPU):
        r, k = rate(H, 0.01, 50, np.tile(S, (4, 1)), False, th)
        rows.append(("B3", "C2 n=1008", "edge-list", th, f"{r:,.0f} syndromes/s ({k} decoded)"))
    # ---- C3: n=16384, realistic and full-50
    H = ldpc.codes.parity_check_csc(16384, 8, 4)
    for per, tag in ((0.02, "per 0.02 (realistic)"), (0.10, "per 0.10 (full-50)")):
        S = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(16384, 4096, per, seed=2))
        r, k = rate(H, per, 50, S[:64], True, 1, budget_s=20)
        rows.append(("B4", f"C3 n=16384, {tag}", "reference-faithful (2 GiB scratch)", 1, f"{r:,.1f} syndromes/s ({k} decoded, not extrapolated)"))
        for th in (1, NCPU):
            r, k = rate(H, per, 50, S, False, th, budget_s=20)
            rows.append(("B5", f"C3 n=16384, {tag}", "edge-list", th, f"{r:,.1f} syndromes/s ({k} decoded)"))
    # ---- C5: BB-72, BP then OSD-0 on what BP leaves unconverged
    HX = ldpc.codes.bivariate_bicycle_72_12_6()[0]
    M = sp.csc_matrix(HX); M.sort_indices()
    S = ldpc.codes.syndromes_of(M, ldpc.codes.random_errors(72, 200000, 0.005, seed=3))
    Hd = HX.astype(np.uint8)

    def bposd_rate(threads):
        def work(chunk):
            oc = BPOracle(csc=(M.indptr, M.indices), shape=M.shape, per=0.005, max_iters=50)
            n_osd = 0
            for b in range(chunk.shape[0]):
                err, conv = oc.decode(chunk[b])
                if not conv:
                    osd_oracle_postprocess(Hd, chunk[b], err.astype(np.uint8), oc.log_probabs, 0)
                    n_osd += 1
            return chunk.shape[0], n_osd
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(threads) as ex:
            res = list(ex.map(work, np.array_split(S, threads)))
        dt = time.perf_counter() - t0
        return sum(r[0] for r in res) / dt, sum(r[1] for r in res)

    for th in (1, NCPU):
        r, nosd = bposd_rate(th)
        rows.append(("B6", "C5 BB-72, per 0.005, BP + OSD-0", "edge-list BP + dense OSD oracle", th,
                     f"{r:,.0f} syndromes/s (200,000 decoded, {nosd} needed OSD; python loop overhead included)"))
    print(f"host: {os.cpu_count()} logical CPUs, {NCPU} used")
    print("| id | config | CPU mode | threads | measured |")
    print("|---|---|---|---|---|")
    for r in rows:
        print("| " + " | ".join(str(x) for x in r) + " |")


if __name__ == "__main__":
    main()
