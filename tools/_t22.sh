export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_default_mode.py tests/test_large_dims.py -m gpu -x -q -k "vlsac or default or soak or pipelined" > gpurun_out/t22_tests.log 2>&1 || { tail -n 30 gpurun_out/t22_tests.log; exit 1; }
tail -n 3 gpurun_out/t22_tests.log
bash tools/_ab_env.sh vlsac_halfcheetah_f256_b256 3000 "-" "RLREP_HV_RECORD=1"
