export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
L=$R/rlrep_amd/lib
bash tools/_ab_env.sh ctrlsac_halfcheetah_f2048_b256 500 "RLREP_LIB=$L/librlrep_hip_bd0bc5b.so" "-" "RLREP_LIB=$L/librlrep_hip_al64.so" "RLREP_LIB=$L/librlrep_hip_al256.so" | sed "s#$L/librlrep_hip_##"
