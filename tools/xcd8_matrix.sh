#!/bin/bash
# Does an eighth team pay once the message slots are small enough?  Persistent teams on 7 against 8 XCDs for (4,8)-regular
# codes of 24 / 28 / 32 MiB a slot and a (3,6) code of 24 MiB, all 50 iterations, 256 tiles; budget raised so that 8 slots
# are always allowed.  (experiments build: the knobs are read from the environment)
L=gpurun_out/xcd8_matrix.log
: > $L
for cfg in "N=16384" "N=14336" "N=12288" "N=16380,WR=6,WC=3" "N=16000,WR=10,WC=5"; do
  for x in 7 8; do
    for rows in 1 0; do
      echo "== $cfg XCDS=$x ROWS=$rows" >> $L
      env ${cfg//,/ } LDPC_TEAM_XCDS=$x LDPC_TEAM_ROWS=$rows LDPC_TEAM_CACHE_MIB=400 BATCHES=${BATCHES:-16384} timeout -k 10 120 python tools/team_mall_probe.py >> $L 2>&1 || { echo "FAILED" >> $L; }
    done
  done
done
grep -v amdgpu.ids $L
