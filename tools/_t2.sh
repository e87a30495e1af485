export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 -m pytest tests -x -q -m gpu -k "vlsac or deferred or default_mode or checkpoint or cross or soak or agree" > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -n 12 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
bash tools/_ab_env.sh vlsac_halfcheetah_f256_b256 2000 "-" "RLREP_NO_CHAIN_NEXT=1"
