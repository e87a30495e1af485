export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
bash tools/_ab_env.sh ctrlsac_halfcheetah_f2048_b256 600 "-" "RLREP_X3S_ALIGNED_ONLY=1" "RLREP_GEMM16_NO_FAST=1"
