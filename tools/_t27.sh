export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/t27_tests.log 2>&1 || { tail -n 30 gpurun_out/t27_tests.log; exit 1; }
tail -n 3 gpurun_out/t27_tests.log
bash tools/_ab_env.sh vlsac_halfcheetah_f256_b256 3000 "-" "RLREP_GEMM16_NO_FAST=1"
