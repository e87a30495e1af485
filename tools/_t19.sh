export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_large_dims.py tests/test_default_mode.py -m gpu -x -q -k "speder or ragged or soak or large" > gpurun_out/t19_tests.log 2>&1 || { tail -n 30 gpurun_out/t19_tests.log; exit 1; }
tail -n 3 gpurun_out/t19_tests.log
bash tools/_ab_env.sh spedersac_ant_f512_b1024 600 "-" "RLREP_NO_FOLD_DWFIN=1"
