#!/usr/bin/env python3
"""Randomised parity fuzz (GPU box): random Tanner graphs, channel probabilities, iteration caps,
batch sizes and kernel/geometry options, every result compared with the CPU oracle (hard decisions,
flags, iteration counts bit-exact; LLRs <= 1e-5, +-Inf exact).  Usage: fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MIN_ROWS_DEFAULT = os.environ.get("LDPC_TEAM_MIN_ROWS", "1")   # kernel_variant 4: teams of workgroups also on these tiny graphs
os.environ["LDPC_TEAM_MIN_ROWS"] = MIN_ROWS_DEFAULT            # (read at create since round 4: the `big` legs below drop it)
import ldpcdecoders_jl_amd as ldpc  # noqa: E402
from oracle import BPOracle, BPOTSOracle  # noqa: E402

# FUZZ_DRY=1: draw the cases and print them (index, graph, knobs) without touching the GPU or the oracle -- the draws do not
# depend on any result, so this lists what a run with the same seed decodes; FUZZ_FROM / FUZZ_TO: only decode the cases
# with these indices (the others are drawn and skipped); FUZZ_VERBOSE=1: one line per case BEFORE it runs (a hang names itself)
DRY = os.environ.get("FUZZ_DRY") == "1"
FROM, TO = int(os.environ.get("FUZZ_FROM", "0")), int(os.environ.get("FUZZ_TO", str(1 << 60)))
VERBOSE = os.environ.get("FUZZ_VERBOSE") == "1"
MAXCASES = int(os.environ.get("FUZZ_CASES", str(1 << 60)))
# FUZZ_ORACLE_ONLY=1: no GPU at all -- time the CPU oracles of the cases FUZZ_FROM ... FUZZ_TO (every leg over 3 s is reported, and
# with FUZZ_VERBOSE=1 every leg's seconds); FUZZ_OTS_CAP: syndromes the BP-OTS leg takes at most (600; 0 = the whole batch, as
# the runs of round 3 before the cap did)
ORACLE_ONLY = os.environ.get("FUZZ_ORACLE_ONLY") == "1"
OTS_CAP = int(os.environ.get("FUZZ_OTS_CAP", "600"))
if not (DRY or ORACLE_ONLY):
    # every host-side wait of the library is bounded (host_wait.hpp): on these small graphs no leg needs more than seconds,
    # so a stall names itself after 40 s (LDPC_ERR_HIP, "<which wait>: the device did not get there ...") instead of
    # sitting silent until the runner's limit
    for exp_build in (False, True):
        ldpc._capi.check(ldpc._capi.lib(exp_build).ldpc_set_wait_limit_ms(int(os.environ.get("FUZZ_WAIT_LIMIT_MS", "40000"))), ldpc._capi.lib(exp_build))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0, cases, decoded = time.time(), 0, 0
last_note = t0
while time.time() - t0 < budget and cases < MAXCASES:
    in_range = FROM <= cases <= TO
    skip = DRY or ORACLE_ONLY or not in_range
    kind = rng.integers(0, 5)
    mid = big = False
    if kind == 0:      # Gallager regular
        wr, wc = int(rng.choice([4, 6, 8, 10])), int(rng.choice([2, 3, 4, 5]))
        n = wr * int(rng.integers(4, 60))
        H = ldpc.codes.parity_check_csc(n, wr, wc, seed=int(rng.integers(1 << 30)))
    elif kind == 4:    # regular with a rows-on-chip instantiation, large enough for the waves of a small team to own chunks
        wr, wc = int(rng.integers(6, 11)), int(rng.integers(3, 6))       # (round 4: every pair of check degree 6 ... 10 x bit degree 3 ... 5)
        big = rng.random() < 0.12
        irr = (not big) and rng.random() < 0.25
        if irr:        # irregular, large enough for small teams: whole checks in the LDS of their owners (IRR), a few checks of 17 ... 32
            #            edges (two halves of the 16-wide code) and now and then one beyond 32 (the O(deg^2) path)
            n = int(rng.integers(1200, 4000)); s_ = n // 2
            rows_, cols_ = [], []
            for j in range(n):
                for i in rng.choice(s_, int(rng.integers(2, 6)), replace=False):
                    rows_.append(int(i)); cols_.append(j)
            for i in rng.choice(s_, int(rng.integers(0, 6)), replace=False):
                for j in rng.choice(n, int(rng.integers(10, 24)), replace=False):
                    rows_.append(int(i)); cols_.append(int(j))
            if rng.random() < 0.4:
                i = int(rng.integers(0, s_))
                for j in rng.choice(n, 40, replace=False):
                    rows_.append(i); cols_.append(int(j))
            H = sp.csc_matrix((np.ones(len(rows_), dtype=np.uint8), (rows_, cols_)), shape=(s_, n))
            H.sum_duplicates(); H.data[:] = 1
        elif big:      # >= 35,200 message rows: the plan's own rules apply (no LDPC_TEAM_MIN_ROWS): the one team of an XCD takes all its CUs
            n = wr * int(rng.integers(36000 // (wr * wc) + 1, 50000 // (wr * wc)))
            n -= n % (wr * 4)   # (so that n * wc / wr is whole and the Gallager blocks divide)
        else:
            n = wr * int(rng.integers(150, 500))
        if not irr:
            H = ldpc.codes.parity_check_csc(n, wr, wc, seed=int(rng.integers(1 << 30)))
        mid = True
    else:              # irregular random, with empty and heavy nodes now and then
        s, n = int(rng.integers(1, 80)), int(rng.integers(1, 160))
        A = (rng.random((s, n)) < rng.uniform(0.02, 0.25)).astype(np.uint8)
        if rng.random() < 0.3:
            A[rng.integers(0, s), :] = 0
        if rng.random() < 0.3:
            A[:, rng.integers(0, n)] = 0
        if rng.random() < 0.2:
            A[rng.integers(0, s), : min(n, 40)] = 1
        if rng.random() < 0.2 and s > 20:
            A[:20, rng.integers(0, n)] = 1
        H = sp.csc_matrix(A)
    H.sort_indices()
    s, n = H.shape
    per = float(rng.choice([1e-6, 0.005, 0.02, 0.05, 0.1, 0.3, 0.5, 0.9]))
    iters = int(rng.choice([1, 2, 3, 7, 20, 50]))
    B = int(rng.choice([1, 2, 63, 64, 65, 130, 400, 1500, 5000]))
    if mid:
        B = int(rng.choice([64, 130, 400, 700]))
        iters = int(rng.choice([3, 7, 20]))
    if big:
        B = int(rng.choice([64, 130, 400]))
        iters = int(rng.choice([3, 7]))
    if rng.random() < 0.5:
        E = (rng.random((B, n)) < min(per * rng.uniform(0.5, 3), 0.5)).astype(np.uint8)
        syn = ldpc.codes.syndromes_of(H, E)
    else:
        syn = rng.integers(0, 2, (B, s)).astype(np.uint8)
    if rng.random() < 0.1 and s > 0:
        syn[rng.integers(0, B), rng.integers(0, s)] = rng.integers(2, 5)
    if DRY or VERBOSE:
        print(f"case {cases}: kind {int(kind)} shape {H.shape} nnz {H.nnz} per {per} iters {iters} B {B}", flush=True)
    if not skip or (ORACLE_ONLY and in_range):
        t_or = time.time()
        oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=iters)
        oerr, oconv, ollr, oits = oc.batchdecode(syn)
        if ORACLE_ONLY and VERBOSE:
            print(f"   BP oracle {time.time() - t_or:.2f} s", flush=True)
        if time.time() - t_or > 3.0:
            print(f"SLOW ORACLE {time.time() - t_or:.1f} s: seed {seed} case {cases} kind {int(kind)} shape {H.shape} nnz {H.nnz} per {per} iters {iters} B {B}", flush=True)
    for variant in (0, 1, 3, 4):
        # node kernel: messages in LDS (default for these small graphs), split between LDS and the global slot
        # at a random point (hybrid), or all in the global slot
        for k in ("LDPC_NODE_MSG_LDS", "LDPC_NODE_HYBRID", "LDPC_NODE_LDS_ROOM"):
            os.environ.pop(k, None)
        place = int(rng.integers(0, 3)) if variant == 3 else 0
        if place >= 1:
            os.environ["LDPC_NODE_MSG_LDS"] = "0"
        if place == 1:
            os.environ["LDPC_NODE_LDS_ROOM"] = str(int(rng.integers(8, max(9, 8 * H.nnz))))
        if place == 2:
            os.environ["LDPC_NODE_HYBRID"] = "0"
        # hand-off levels (read at create): thresholds of fresh / level-1 tiles, tiny level capacities (levels fill
        # up and tiles must carry on), how many stragglers the node kernel finishes
        for k in ("LDPC_DEFER_T0", "LDPC_DEFER_T1", "LDPC_DEFER_CAP_TILES", "LDPC_NODE_TAKE_MAX", "LDPC_TEAM_CACHE_KIB",
                  "LDPC_TEAM_DYNAMIC", "LDPC_TEAM_PAIRS", "LDPC_TEAM_ROWS", "LDPC_TEAM_AHEAD", "LDPC_TEAM_REGS", "LDPC_TEAM_STATIC",
                  "LDPC_TEAM_MAX", "LDPC_TEAM_CONCENTRATE", "LDPC_TEAM_FLIP", "LDPC_TEAM_AHEAD_FROM", "LDPC_TEAM_WIDE", "LDPC_TEAM_PRE",
                  "LDPC_TEAM_STRAYS_LAST"):
            os.environ.pop(k, None)
        # running ahead (two team barriers an iteration on quiet tiles), rows in the waves' accumulator registers and how
        # much of a member's share its waves own by right; few members on the mid-size graphs so that every wave owns chunks
        if rng.random() < 0.7:
            os.environ["LDPC_TEAM_AHEAD"] = str(int(rng.choice([0, 1, 20, 32, 64])))
        if rng.random() < 0.4:
            os.environ["LDPC_TEAM_AHEAD_FROM"] = str(int(rng.integers(1, 4)))
        if rng.random() < 0.5:
            os.environ["LDPC_TEAM_REGS"] = str(int(rng.choice([0, 5, 32])))
        if rng.random() < 0.5:
            os.environ["LDPC_TEAM_STATIC"] = str(int(rng.choice([1, 2, 3, 4])))
        wide = mid and B >= 256 and rng.random() < 0.2
        if wide:       # a few persistent teams over ALL XCDs with rows on chip (the plan of graphs beyond n = 24576), forced on these
            os.environ["LDPC_TEAM_WIDE"] = str(int(rng.choice([1, 2, 4, 7])))
        elif mid and not big:
            os.environ["LDPC_TEAM_MAX"] = str(int(rng.choice([3, 4, 6, 8])))
        if big:
            os.environ.pop("LDPC_TEAM_MIN_ROWS", None)
        else:
            os.environ["LDPC_TEAM_MIN_ROWS"] = MIN_ROWS_DEFAULT
        # teams (read at create): a small cache budget makes them persistent on these small graphs (a team takes tile
        # after tile in its own slot); how a member's waves share its chunks; nodes loaded in pairs or singly
        if rng.random() < 0.6:
            os.environ["LDPC_TEAM_CACHE_KIB"] = str(int(rng.choice([1, 8 * max(1, H.nnz) // 2, 16 * max(1, H.nnz)])))
        if rng.random() < 0.5:
            os.environ["LDPC_TEAM_DYNAMIC"] = str(int(rng.integers(0, 2)))
        if rng.random() < 0.5:
            os.environ["LDPC_TEAM_PAIRS"] = str(int(rng.integers(0, 4)))   # bit 0: two nodes at a time, bit 1: four bits
        # which rows end up on chip (0: most room, 1: whole checks of the first block, 2: only whole checks), and the
        # order in which the upper waves walk their static chunks
        if rng.random() < 0.5:
            os.environ["LDPC_TEAM_CONCENTRATE"] = str(int(rng.integers(0, 3)))
        if rng.random() < 0.5:
            os.environ["LDPC_TEAM_FLIP"] = str(int(rng.integers(0, 4)))
        # on-chip checks updated between arriving at the barrier behind the variable sweep and waiting at it (0 = a plain barrier);
        # stray bits at the end of a member's positions or dealt by number
        if rng.random() < 0.5:
            os.environ["LDPC_TEAM_PRE"] = str(int(rng.integers(0, 5)))
        if rng.random() < 0.3:
            os.environ["LDPC_TEAM_STRAYS_LAST"] = str(int(rng.integers(0, 2)))
        if rng.random() < 0.3:
            os.environ["LDPC_TEAM_ROWS"] = "0"      # regular graphs: no rows in LDS / registers
        if rng.random() < 0.7:
            os.environ["LDPC_DEFER_T0"] = str(int(rng.choice([4, 16, 32, 48])))
            os.environ["LDPC_DEFER_T1"] = str(int(rng.choice([0, 8, 16, 40])))
        if rng.random() < 0.3:
            os.environ["LDPC_DEFER_CAP_TILES"] = str(int(rng.choice([1, 2, 5])))
        if rng.random() < 0.5:
            os.environ["LDPC_NODE_TAKE_MAX"] = str(int(rng.choice([0, 3, 40, 100000])))
        kw = dict(kernel_variant=variant, waves_per_tile=0 if variant == 4 else int(rng.choice([0, 4, 8, 16])),
                  defer_threshold=int(rng.choice([0, -1, 4, 40])))
        if variant != 4 and rng.random() < 0.3:
            kw["resident_tiles"] = int(rng.integers(1, 5))
        want_llr = bool(rng.random() < 0.5)
        if rng.random() < 0.3:
            kw["llr_exact"] = True    # LLRs from the full posterior odds (default: their upper 32 bits; ldpc_bp_options.llr_exact)
        knobs = {k[5:]: v for k, v in os.environ.items() if k.startswith("LDPC_") and not k.startswith("LDPC_MI355X")}
        if DRY or VERBOSE:
            print(f"   variant {variant} {kw} llr={want_llr} {knobs}", flush=True)
        if not skip:   # what is about to run, where a run that stalls leaves it behind (gpurun merges gpurun_out/ back)
            try:
                with open("gpurun_out/fuzz_current.txt", "w") as fh:
                    fh.write(f"seed {seed} case {cases} after {time.time() - t0:.0f} s: kind {int(kind)} shape {H.shape} nnz {H.nnz} per {per} "
                             f"iters {iters} B {B}\n   variant {variant} {kw} llr={want_llr} {knobs}\n")
            except OSError:
                pass
        if skip:
            continue
        t_leg = time.time()
        dec = ldpc.BeliefPropagationDecoder(H, per, iters, **kw)
        err, conv, llr, its = dec.decode_batch_host(syn, want_llr=want_llr, want_iters=True)
        ok = np.array_equal(err, oerr) and np.array_equal(conv, oconv) and np.array_equal(its, oits)
        if ok and want_llr:
            fin = np.isfinite(ollr)
            tol = 1e-9 if kw.get("llr_exact") else 1e-6    # (two libms / the cut of the odds to 21 significant bits; BASELINE.json asks for 1e-5)
            ok = np.array_equal(llr[~fin], ollr[~fin]) and (not fin.any() or np.max(np.abs(llr[fin] - ollr[fin])) <= tol)
        if not ok:
            np.savez("gpurun_out/fuzz_failure.npz", colptr=H.indptr, rowval=H.indices, shape=np.array(H.shape), per=per,
                     iters=iters, syn=syn)
            print(f"MISMATCH case {cases}: shape {H.shape} nnz {H.nnz} per {per} iters {iters} B {B} {kw} llr={want_llr}")
            sys.exit(1)
        dec.close()
        if time.time() - t_leg > 3.0:   # a leg that took seconds on these small graphs: say which (timeouts inside the library?)
            msg = (f"SLOW {time.time() - t_leg:.1f} s: seed {seed} case {cases} kind {int(kind)} shape {H.shape} nnz {H.nnz} per {per} iters {iters} B {B} "
                   f"variant {variant} {kw} llr={want_llr} {knobs} last_kernel {getattr(dec, '_last_kernel_seen', '?')}")
            print(msg, flush=True)
            try:
                with open("gpurun_out/fuzz_slow.txt", "a") as fh:
                    fh.write(msg + "\n")
            except OSError:
                pass
        decoded += B
    if kind == 1 and H.nnz > 0:      # (any degree: nodes beyond 32 / 16 edges take the unlimited kernel)
        T, C = int(rng.choice([2, 3, 9])), float(rng.choice([1.0, 2.0, 3.0]))
        # (the CPU oracle of BP-OTS re-runs with biases and takes minutes for 5,000 syndromes x 50 iterations -- what looked
        #  like a stalled run in round 3 was this: the leg decodes the first 600 syndromes of the batch at most)
        syn_ots = syn[:OTS_CAP] if OTS_CAP > 0 else syn
        pp = max(per, 1e-3) if per < 0.9 else 0.3
        os.environ.pop("LDPC_BPOTS_FORCE_NODE", None)
        if rng.random() < 0.6:      # the node-parallel kernel (graphs beyond the LDS) or the unlimited one on a small graph
            os.environ["LDPC_BPOTS_FORCE_NODE"] = str(int(rng.integers(1, 3)))
        if DRY or VERBOSE:
            print(f"   BP-OTS T {T} C {C} per {pp} force {os.environ.get('LDPC_BPOTS_FORCE_NODE')}", flush=True)
        if ORACLE_ONLY and in_range:
            t_or = time.time()
            BPOTSOracle((H.indptr, H.indices), H.shape, pp, iters, T, C).batchdecode(syn_ots)
            if VERBOSE:
                print(f"   BP-OTS oracle {time.time() - t_or:.2f} s ({len(syn_ots)} syndromes)", flush=True)
            if time.time() - t_or > 3.0:
                print(f"SLOW BP-OTS ORACLE {time.time() - t_or:.1f} s: seed {seed} case {cases} shape {H.shape} nnz {H.nnz} per {pp} iters {iters} B {B} T {T} C {C}", flush=True)
        if not skip:
            try:
                with open("gpurun_out/fuzz_current.txt", "w") as fh:
                    fh.write(f"seed {seed} case {cases} after {time.time() - t0:.0f} s: BP-OTS shape {H.shape} nnz {H.nnz} per {pp} iters {iters} B {B} "
                             f"T {T} C {C} force {os.environ.get('LDPC_BPOTS_FORCE_NODE')}\n")
            except OSError:
                pass
            t_or = time.time()
            oe, ocv, oi = BPOTSOracle((H.indptr, H.indices), H.shape, pp, iters, T, C).batchdecode(syn_ots)
            if time.time() - t_or > 3.0:
                print(f"SLOW BP-OTS ORACLE {time.time() - t_or:.1f} s: seed {seed} case {cases} shape {H.shape} nnz {H.nnz} per {pp} iters {iters} B {B} T {T} C {C}", flush=True)
            d2 = ldpc.BPOTSDecoder(H, pp, iters, T=T, C=C)
            e2, c2, i2 = d2.decode_batch_host(syn_ots)
            if not (np.array_equal(e2, oe) and np.array_equal(c2, ocv) and np.array_equal(i2, oi)):
                print(f"BP-OTS MISMATCH case {cases}: shape {H.shape} nnz {H.nnz} per {pp} iters {iters} B {B} T {T} C {C}")
                sys.exit(1)
            d2.close()
    cases += 1
    if time.time() - last_note > 60:   # a silent GPU job looks hung to the runner
        last_note = time.time()
        print(f"... {cases} cases, {decoded} syndromes, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz ok: {cases} random cases, {decoded} syndromes decoded on the GPU in {time.time() - t0:.0f} s, seed {seed}")
