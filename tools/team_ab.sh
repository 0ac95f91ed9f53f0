#!/bin/bash
# medium batches through the team kernel (C3 code), ms per call by batch size and error rate; with an argument:
# A/B against another build of the library (LDPC_MI355X_LIB), alternating, same box
run() {
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ldpcdecoders_jl_amd as ldpc
n = 16384
H = ldpc.codes.parity_check_csc(n, 8, 4)
for per in (0.10, 0.02):
    row = []
    for B in (1, 64, 256, 1024, 2048, 4096, 8192):
        syn = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, per, seed=B))).cuda()
        dec = ldpc.BeliefPropagationDecoder(H, per, 50)
        err = torch.empty((B, n), dtype=torch.uint8, device="cuda"); conv = torch.empty(B, dtype=torch.uint8, device="cuda")
        for _ in range(2): dec.decode_batch_device(syn, err, conv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps): dec.decode_batch_device(syn, err, conv)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        row.append(f"{B}: {ms:.2f} ms (k{dec.info().last_kernel} G{dec.info().last_team_size})")
        dec.close()
    print(f"per {per}: " + "  ".join(row))
PY
}
if [ -n "$1" ]; then
  for rep in 1 2; do echo "== this build"; run; echo "== $1"; LDPC_MI355X_LIB=$PWD/$1 run; done
else
  run
fi
