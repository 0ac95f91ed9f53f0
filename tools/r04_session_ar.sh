#!/bin/bash
# arrival on a counter per XCD for teams over several XCDs (-DLDPC_TEAM_HIER=1, experiments-build variant) against the experiments build: wide teams,
# single decode; parity of the variant on the team tests first
C=$PWD/ldpcdecoders.jl_amd/csrc
V="LDPC_MI355X_LIB=$C/libldpc_v_hier.so LDPC_MI355X_EXP_LIB=$C/libldpc_v_hier.so"
tools/gpu_session.sh r04ar \
  500 "$V python -m pytest tests -m gpu -x -q -k 'team or wide or single or latency or scatter'" \
  200 "N=65536 MODE=auto_vs_off python tools/wide_teams_probe.py" \
  200 "$V N=65536 MODE=auto_vs_off python tools/wide_teams_probe.py" \
  200 "N=32768 MODE=auto_vs_off python tools/wide_teams_probe.py" \
  200 "$V N=32768 MODE=auto_vs_off python tools/wide_teams_probe.py" \
  200 "LDPC_TEAM_DEBUG=0 CASES=16384:0.02,16384:0.10 AUTO=1 BATCHES=1,256 python tools/smallbatch_probe.py" \
  200 "$V LDPC_TEAM_DEBUG=0 CASES=16384:0.02,16384:0.10 AUTO=1 BATCHES=1,256 python tools/smallbatch_probe.py"
tail -3 gpurun_out/r04ar_1.log
for k in 2 3 4 5 6 7; do grep "^n " gpurun_out/r04ar_$k.log | cut -c1-200; done
