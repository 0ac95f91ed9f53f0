#!/usr/bin/env python3
"""Graphs whose message slot is a large part of the Infinity Cache (n = 65536: 128 MiB a slot, eight one-XCD teams' slots
are four times the cache and the plan hands the large batches to the HBM-streaming tile kernel): do a FEW persistent
teams over ALL XCDs, rows on chip, whose slots fit the cache, beat it?  (LDPC_TEAM_WIDE = teams; experiments build.)
Every configuration must give the tile kernel's results bit for bit.  N, BATCH, ITERS from the environment."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpcdecoders_jl_amd as ldpc

n, batch, iters, per = int(os.environ.get("N", "65536")), int(os.environ.get("BATCH", "16384")), int(os.environ.get("ITERS", "50")), 0.10
H = ldpc.codes.parity_check_csc(n, 8, 4)
S = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, batch, per, seed=3))).cuda()
ref = None
CONFIGS = [("tile kernel", 1, {}), ("auto", 0, {}), ("wide 1", 0, {"LDPC_TEAM_WIDE": "1"}), ("wide 2", 0, {"LDPC_TEAM_WIDE": "2"}),
           ("wide 4", 0, {"LDPC_TEAM_WIDE": "4"}), ("wide 2, no rows", 0, {"LDPC_TEAM_WIDE": "2", "LDPC_TEAM_ROWS": "0"}),
           ("8 one-XCD teams", 0, {"LDPC_TEAM_XCDS": "8", "LDPC_TEAM_CACHE_MIB": "4000"})]
if os.environ.get("MODE") == "auto_vs_off":      # what the plan does by itself against the plan without wide teams
    CONFIGS = [("wide teams off", 0, {"LDPC_TEAM_WIDE": "-1"}), ("auto", 0, {})] + [(f"wide {t}", 0, {"LDPC_TEAM_WIDE": t}) for t in os.environ.get("ALSO", "").split(",") if t]
    if os.environ.get("TILE") == "1":
        CONFIGS.append(("tile kernel", 1, {}))
for name, variant, env in CONFIGS:
    for k in ("LDPC_TEAM_WIDE", "LDPC_TEAM_ROWS", "LDPC_TEAM_XCDS", "LDPC_TEAM_CACHE_MIB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    dec = ldpc.BeliefPropagationDecoder(H, per, iters, kernel_variant=variant, experiments=True)
    err = torch.empty((batch, n), dtype=torch.uint8, device="cuda"); conv = torch.empty(batch, dtype=torch.uint8, device="cuda")
    its = torch.empty(batch, dtype=torch.int32, device="cuda")
    dec.decode_batch_device(S, err, conv, None, its)
    dec.last_status()
    ts = []
    for _ in range(2):
        dec.decode_batch_device(S, err, conv, None, its)
        dec.last_status()
        ts.append(dec.last_timing()[0])
    inf = dec.info()
    ck, vr, rs = dec.phase_ticks()
    nt = (batch + 63) // 64
    same = "reference" if ref is None else ("identical" if all(torch.equal(a, b) for a, b in zip(ref, (err, conv, its))) else "DIFFERENT")
    if ref is None:
        ref = (err.clone(), conv.clone(), its.clone())
    tb = batch * iters * 32.0 * H.nnz / (min(ts) * 1e-3) / 1e12
    print(f"n {n} batch {batch} {name:18s}: kernel {min(ts):8.1f} ms  {tb:5.2f} TB/s algorithmic  k{inf.last_kernel} G{inf.last_team_size} slots {inf.resident_tiles // max(inf.last_team_size, 1)} "
          f"rows on chip {inf.last_rows_on_chip}  per team-iteration: check {ck / nt / iters / 100:6.1f} var {vr / nt / iters / 100:6.1f} rest {rs / nt / iters / 100:6.1f} us  {same}", flush=True)
    dec.close()
    del err, conv, its
