#!/usr/bin/env python3
"""Team kernel (kernel_variant 4) on the C3 code, all 50 iterations: ms per call for a few batch sizes and, from
the kernel's own phase clocks (ldpc_bp_call_phase_ticks), what one member spends per iteration in its check
sweep, its variable sweep and everything else (team barriers + convergence test, waiting included).
BATCHES=64,512,...  LDPC_TEAM_DEBUG=1 additionally prints which XCDs the teams landed on."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpcdecoders_jl_amd as ldpc
n=16384; per=0.10
H = ldpc.codes.parity_check_csc(n, 8, 4)
dec = ldpc.BeliefPropagationDecoder(H, per, 50, kernel_variant=4)
for batch in [int(x) for x in os.environ.get('BATCHES', '64,512,1024,2048').split(',')]:
    S = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, batch, per, seed=3))
    Sd = torch.from_numpy(S).cuda()
    err = torch.empty((batch, n), dtype=torch.uint8, device="cuda"); conv = torch.empty(batch, dtype=torch.uint8, device="cuda")
    for _ in range(2): dec.decode_batch_device(Sd, err, conv)
    torch.cuda.synchronize()
    ts=[]
    for _ in range(4):
        t0=time.perf_counter(); dec.decode_batch_device(Sd, err, conv); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    tk = dec.phase_ticks(0); nt = (batch + 63) // 64
    print(batch, "ms", [round(t*1e3,2) for t in ts], "per team-iteration us: check %.1f var %.1f rest(barriers+test) %.1f" % tuple(x / 100.0 / nt / 50 for x in tk), flush=True)
