#!/usr/bin/env python3
"""Team kernel on a few tiles of the C3 code (n 16384, full 50 iterations at per 0.10): how fast do the sweeps run when
the tiles in flight fit the 256 MiB Infinity Cache?  Prints ms per call and the algorithmic rate (32 nnz B per syndrome
and iteration).  Geometry through LDPC_TEAM_MAX / LDPC_TEAM_MIN_ROWS / LDPC_TEAM_PER_CU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpcdecoders_jl_amd as ldpc

n, per = int(os.environ.get("N", "16384")), 0.10
H = ldpc.codes.parity_check_csc(n, int(os.environ.get("WR", "8")), int(os.environ.get("WC", "4")))
variant = int(os.environ.get("VARIANT", "4"))
dec = ldpc.BeliefPropagationDecoder(H, per, 50, kernel_variant=variant)
for batch in [int(x) for x in os.environ.get("BATCHES", "64,256,384,512,768,1024,2048").split(",")]:
    E = ldpc.codes.random_errors(n, batch, per, seed=3)
    S = ldpc.codes.syndromes_of(H, E)
    Sd = torch.from_numpy(np.ascontiguousarray(S)).cuda()
    err = torch.empty((batch, n), dtype=torch.uint8, device="cuda")
    conv = torch.empty(batch, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        dec.decode_batch_device(Sd, err, conv)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); dec.decode_batch_device(Sd, err, conv); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) * 1e3
    kernel_ms, total_ms, sum_iters = dec.last_timing()
    iters = sum_iters / batch
    tb = batch * iters * 32.0 * H.nnz / (kernel_ms * 1e-3) / 1e12
    ck, vr, rs = dec.phase_ticks()
    nt = (batch + 63) // 64
    inf = dec.info()
    ph = f"per team and iteration: check {ck / nt / iters / 100:6.1f} us  var {vr / nt / iters / 100:6.1f} us  barriers+test {rs / nt / iters / 100:6.1f} us  (k{inf.last_kernel} G{inf.last_team_size} slots {inf.resident_tiles // max(inf.last_team_size, 1)} lds_rows {inf.last_lds_rows})"
    if os.environ.get("DIAG3"):   # library built with -DLDPC_TEAM_DIAG=3: max / min over the members' own sweep times
        ph = f"slowest member: check {ck / iters / 100:6.1f} us  var {vr / iters / 100:6.1f} us per iteration; fastest member's check {((1 << 64) - 1 - rs) / iters / 100:6.1f} us  (k{inf.last_kernel} G{inf.last_team_size})"
    print(f"batch {batch:5d} ({(batch + 63) // 64:3d} tiles): wall {ms:8.3f} ms  kernel {kernel_ms:8.3f} ms  mean iterations {iters:5.1f}  {tb:5.2f} TB/s algorithmic  {ph}", flush=True)
dec.close()
