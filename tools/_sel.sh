export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "speder" > gpurun_out/sel_tests.log 2>&1 || { tail -n 30 gpurun_out/sel_tests.log; exit 1; }
tail -n 2 gpurun_out/sel_tests.log
bash tools/_ab_env.sh spedersac_ant_f512_b1024 600 "-" "RLREP_NO_FOLD_THETA=1"
