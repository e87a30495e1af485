export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 -m pytest tests -x -q -m gpu -k "vlsac or deferred or default_mode or checkpoint or cross or soak" > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
bash tools/_ab_env.sh vlsac_halfcheetah_f256_b256 2000 "-" "RLREP_NO_FOLD_NCDW=1" "RLREP_NO_MANAGED_IMAGES=1" "RLREP_NO_FOLD_NCDW=1 RLREP_NO_MANAGED_IMAGES=1"
