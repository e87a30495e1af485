export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_default_mode.py -m gpu -x -q -k "vlsac or default or soak or pipelined or chained or decoder" > gpurun_out/t25_tests.log 2>&1 || { tail -n 30 gpurun_out/t25_tests.log; exit 1; }
tail -n 3 gpurun_out/t25_tests.log
bash tools/_ab_env.sh vlsac_halfcheetah_f256_b256 3000 "-" "RLREP_LIB=$R/rlrep_amd/lib/librlrep_hip_prev.so"
RLREP_LIB=$R/rlrep_amd/lib/librlrep_hip_tim.so timeout -k 10 300 python3 tools/exp/gemm_timeline.py > gpurun_out/gemm_timeline2.txt 2>&1
