P=tools/team_mall_probe.py; L=gpurun_out/team_lds.log; : > $L
D3=$PWD/ldpcdecoders.jl_amd/csrc/libldpc_diag3.so
run() { echo "== $*" >> $L; env "$@" BATCHES=512 timeout -k 10 100 python $P >> $L 2>&1 || exit 1; }
run X=0
run LDPC_TEAM_LDS_KIB=96
run LDPC_TEAM_LDS_KIB=60 LDPC_TEAM_NO_MARGIN=1 LDPC_TEAM_MAX=64 LDPC_TEAM_MIN_ROWS=1024
run LDPC_TEAM_NO_MARGIN=1 LDPC_TEAM_PER_CU=3 LDPC_TEAM_MAX=64 LDPC_TEAM_MIN_ROWS=1024
run DIAG3=1 LDPC_MI355X_LIB=$D3
run DIAG3=1 LDPC_MI355X_LIB=$D3 LDPC_TEAM_LDS_KIB=96
run DIAG3=1 LDPC_MI355X_LIB=$D3 LDPC_TEAM_LDS_KIB=60 LDPC_TEAM_NO_MARGIN=1 LDPC_TEAM_MAX=64 LDPC_TEAM_MIN_ROWS=1024
grep -v amdgpu.ids $L
