#!/bin/bash
# wide teams (members over all XCDs: a barrier costs ~20 us there) with 0 / 2 / 4 chunks of on-chip checks in its shadow
S=""
for n in 65536 32768; do for pre in 0 2 4; do S="$S 200 \"N=$n MODE=auto_vs_off LDPC_TEAM_PRE=$pre python tools/wide_teams_probe.py\""; done; done
eval tools/gpu_session.sh r04am $S
for k in 1 2 3 4 5 6; do grep "^n " gpurun_out/r04am_$k.log | cut -c1-200; done
