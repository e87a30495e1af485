export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
bash tools/_ab_env.sh ctrlsac_halfcheetah_f2048_b256 600 "-" "RLREP_LIB=$R/rlrep_amd/lib/librlrep_hip_old.so"
