export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
L=$R/rlrep_amd/lib
timeout -k 10 600 python3 -m pytest tests/test_large_dims.py tests/test_gemm_engines.py -m gpu -x -q > gpurun_out/t37_tests.log 2>&1 || { tail -n 30 gpurun_out/t37_tests.log; exit 1; }
tail -n 2 gpurun_out/t37_tests.log
bash tools/_ab_env.sh ctrlsac_halfcheetah_f2048_b256 500 "RLREP_LIB=$L/librlrep_hip_head.so" "-" | sed "s#$L/librlrep_hip_##"
bash tools/_ab_env.sh spedersac_ant_f512_b1024 500 "RLREP_LIB=$L/librlrep_hip_head.so" "-" | sed "s#$L/librlrep_hip_##"
