export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
bash tools/_ab_env.sh vlsac_halfcheetah_f256_b256 3000 "-" "RLREP_LIB=$R/rlrep_amd/lib/librlrep_hip_prev.so"
