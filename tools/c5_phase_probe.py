#!/usr/bin/env python3
"""BASELINE configs[4] (BB-72, 2^20 syndromes, BP + OSD-0 on the host): where the time of
BeliefPropagationOSDDecoder.batchdecode_device goes -- allocation, BP on the GPU, finding the unconverged
syndromes, copying them to the host, OSD, copying the estimates back."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpcdecoders_jl_amd as ldpc
HX,_ = ldpc.codes.bivariate_bicycle_72_12_6()
B=1<<20
dec = ldpc.BeliefPropagationOSDDecoder(HX, 0.005, 50, osd_order=0)
E = ldpc.codes.random_errors(72, B, 0.005, seed=9)
S = ldpc.codes.syndromes_of(HX, E)
syn = torch.from_numpy(S).cuda()
def sync(): torch.cuda.synchronize()
for _ in range(3): dec.batchdecode_device(syn)
sync()
bp = dec.bp_decoder
for rep in range(3):
    t=[time.perf_counter()]
    err = torch.empty((B, bp.n), dtype=torch.uint8, device="cuda"); conv = torch.empty(B, dtype=torch.uint8, device="cuda"); llr = torch.empty((B, bp.n), dtype=torch.float64, device="cuda")
    sync(); t.append(time.perf_counter())
    bp.decode_batch_device(syn, err, conv, llr, None); sync(); t.append(time.perf_counter())
    idx = torch.nonzero(conv == 0, as_tuple=False).flatten(); k=int(idx.numel()); sync(); t.append(time.perf_counter())
    a,b,c = syn[idx].cpu().numpy(), err[idx].cpu().numpy(), llr[idx].cpu().numpy(); t.append(time.perf_counter())
    out = dec._osd.postprocess(a,b,c, nthreads=0); t.append(time.perf_counter())
    err[idx] = torch.from_numpy(out).to("cuda"); sync(); t.append(time.perf_counter())
    print(k, ["%.3f"%((t[i+1]-t[i])*1e3) for i in range(len(t)-1)], "total %.3f ms"%((t[-1]-t[0])*1e3))
    del err, conv, llr
t0=time.perf_counter()
for _ in range(5): dec.batchdecode_device(syn)
sync(); print("batchdecode_device avg %.3f ms"%((time.perf_counter()-t0)/5*1e3))
